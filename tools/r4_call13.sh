#!/bin/bash
# round 4, GPU call 13: L1 <-> L2 request counters of the gather / scatter kernels of the headline step, calibrated on the
# random-gather microbenchmark (one 128-byte line per lane-load by construction)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_cal -o c -- $R/tools/gather_pair_bench > /dev/null 2>/tmp/e_cal.log
python $R/tools/prof_summary.py $(find /tmp/p_cal -name "*.db" | head -1) $O/c13_cal.json
timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_l2 -o c -- python $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity > /dev/null 2>/tmp/e_l2.log
python $R/tools/prof_summary.py $(find /tmp/p_l2 -name "*.db" | head -1) $O/c13_l2.json
tail -3 /tmp/e_cal.log /tmp/e_l2.log | cut -c1-300
ls -la $O/c13_*
