#!/bin/bash
# round 4, GPU call 9: per-step kernel time of the multi-object and street steps INCLUDING the ATen glue kernels, by the
# difference of two rocprofv3 runs with 8 and 24 timed steps (set-up cancels)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in multi street; do
  for n in 8 24; do
    timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_${cfg}_$n -o s -- python $R/bench.py --config $cfg --steps $n --warmup 4 > $O/c9_${cfg}_$n.json 2>/tmp/e_${cfg}_$n.log
    python $R/tools/prof_summary.py $(find /tmp/p_${cfg}_$n -name "*.db" | head -1) $O/c9_${cfg}_${n}_stats.json
  done
done
ls -la $O/c9_*
