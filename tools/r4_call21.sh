#!/bin/bash
# round 4, GPU call 21: LDS counters of the decoder kernels on the headline step (bank conflicts of the staging / image layouts)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES -d /tmp/p_lds -o c -- python $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity > /dev/null 2>/tmp/e_lds.log
python $R/tools/prof_summary.py $(find /tmp/p_lds -name "*.db" | head -1) $O/c21_lds.json
tail -2 /tmp/e_lds.log | cut -c1-300
