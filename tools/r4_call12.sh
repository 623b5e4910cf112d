#!/bin/bash
# round 4, GPU call 12: kernel stats + idle-gap profile of the drop-in API path (single-threaded autograd)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
export NSIM_FUSED_STEP=0 NSIM_AUTOGRAD_MT=0
cd /tmp && export TMPDIR=/tmp
for n in 16 48; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_api_$n -o s -- python $R/bench.py --steps $n --warmup 16 --no-cpu-baseline --no-variants --no-parity > $O/c12_api_$n.json 2>/tmp/e_$n.log
  python $R/tools/prof_summary.py $(find /tmp/p_api_$n -name "*.db" | head -1) $O/c12_api_${n}_stats.json
done
python $R/tools/gap_profile.py $(find /tmp/p_api_48 -name "*.db" | head -1) $O/c12_api_gaps.json
python $R/tools/prof_diff.py $O/c12_api_16_stats.json 16 $O/c12_api_48_stats.json 48 50
