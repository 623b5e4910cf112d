#!/bin/bash
# round 6, GPU call 5: per-step kernel time of the API path and of the fused chain (difference of two rocprofv3 runs with
# 16 / 48 steps each: set-up kernels cancel, ATen glue stays in), the full-size convergence record, the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_call5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in api fused; do
  [ $mode = api ] && export NSIM_FUSED_STEP=0 || export NSIM_FUSED_STEP=1
  for n in 16 48; do
    rm -rf /tmp/p_${mode}_$n
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_${mode}_$n -o s -- python $R/bench.py --steps $n --warmup 16 --no-cpu-baseline --no-variants --no-parity > /dev/null 2>/tmp/e_${mode}_$n.log
    python $R/tools/prof_summary.py $(find /tmp/p_${mode}_$n -name "*.db" | head -1) $OUT/stats_${mode}_$n.json
  done
  python $R/tools/prof_diff.py $OUT/stats_${mode}_16.json 32 $OUT/stats_${mode}_48.json 64 70 > $OUT/step_kernels_$mode.txt
done
unset NSIM_FUSED_STEP
cd $R
python tools/convergence_run.py > $OUT/convergence.json 2> $OUT/convergence.err
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
head -3 $OUT/step_kernels_api.txt; head -3 $OUT/step_kernels_fused.txt; cat $OUT/convergence.json | tr '\n' ' '; echo
python -c "
import json
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('ms_per_step_p50'), d['variants'].get('api_path_ms'), d['variants'].get('api_path_host_wait_ms'))
"
