#!/bin/bash
# round-3 GPU call 1: distant-path profile, per-level scatter timing, f32 sampling-pass experiment
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-variants --no-parity"
timeout 300 python $R/bench.py --steps 48 --warmup 16 $B > $O/c1_bench_fp16.json 2> $O/c1_bench_fp16.err
NSIM_SAMPLING_PRECISION=f32 timeout 300 python $R/bench.py --steps 48 --warmup 16 $B > $O/c1_bench_sf32.json 2> $O/c1_bench_sf32.err
timeout 300 python $R/tools/scatter_levels.py $O/c1_scatter_levels.json > /dev/null 2> $O/c1_scatter.err
cd $R && NSIM_SAMPLING_PRECISION=f32 timeout 600 python -m pytest tests/test_fullsize_parity.py -x -q -m gpu -k "fp16" > $O/c1_parity_sf32.log 2>&1
mkdir -p $O/c1_parity && cp $O/parity_fullsize_*fp16*.json $O/c1_parity/ 2>/dev/null
cd /tmp
CMD="python $R/bench.py --distant --steps 16 --warmup 8 $B"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_dist -o s -- $CMD > $O/c1_distant_bench.json 2>/tmp/e1.log
python $R/tools/prof_summary.py $(find /tmp/p_dist -name "*.db" | head -1) $O/c1_distant_stats.json
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/p_dsq -o q -- $CMD > /dev/null 2>/tmp/e4.log
python $R/tools/prof_summary.py $(find /tmp/p_dsq -name "*.db" | head -1) $O/c1_distant_pmc_sq.json
tail -c 600 $O/c1_bench_fp16.json; echo; tail -c 600 $O/c1_bench_sf32.json; echo; tail -5 $O/c1_parity_sf32.log
