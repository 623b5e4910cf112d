#!/bin/bash
# round 6 (session 2), GPU call 6: per-step kernel lists of the API path and the fused chain on the current tree (difference of two
# rocprofv3 runs with 16 / 48 timed steps: set-up cancels, ATen glue stays in)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_s2_call6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in api fused; do
  [ $mode = api ] && export NSIM_FUSED_STEP=0 || export NSIM_FUSED_STEP=1
  for n in 16 48; do
    rm -rf /tmp/p_${mode}_$n
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_${mode}_$n -o s -- python $R/bench.py --steps $n --warmup 16 --no-cpu-baseline --no-variants --no-parity > /dev/null 2>/tmp/e_${mode}_$n.log
    python $R/tools/prof_summary.py $(find /tmp/p_${mode}_$n -name "*.db" | head -1) $OUT/stats_${mode}_$n.json
  done
  python $R/tools/prof_diff.py $OUT/stats_${mode}_16.json 32 $OUT/stats_${mode}_48.json 64 90 > $OUT/step_kernels_$mode.txt
done
head -3 $OUT/step_kernels_api.txt; head -3 $OUT/step_kernels_fused.txt
