#!/bin/bash
# round 4, GPU call 17: atomic / read request counters of the STREET step's scatters and gathers
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/p_l2s -o c -- python $R/bench.py --config street --steps 8 --warmup 4 > /dev/null 2>/tmp/e_l2s.log
python $R/tools/prof_summary.py $(find /tmp/p_l2s -name "*.db" | head -1) $O/c17_street_l2.json
ls -la $O/c17_street_l2.json
