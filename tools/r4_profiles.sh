#!/bin/bash
# round 4, profiling call (runs ON THE GPU BOX): (a) run-to-run spread of the permutohedral fp16 full-size parity record (the
# model is pre-trained on the device with float atomics: a different field every process); (b) bench line + rocprofv3 kernel
# stats + gap profile + HBM (FETCH_SIZE / WRITE_SIZE) and SQ / MFMA counter passes of the SAME command, counters in their own
# --kernel-trace-only runs; (c) kernel stats of the street step.  tools/make_profiles.py round4 turns gpurun_out/prof_* into
# the tracked profiles/round4_* files.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for i in $(seq 1 ${PERMUTO_REPS:-6}); do
  timeout 200 python -m pytest tests/test_fullsize_parity.py -m gpu -q -k "permuto_model and fp16" > $O/permuto_fp16_rep$i.log 2>&1
  cp $O/parity_fullsize_permuto_api_fp16_compressed.json $O/permuto_fp16_rep$i.json 2>/dev/null
  tail -1 $O/permuto_fp16_rep$i.log
done
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o s -- $CMD > $O/prof_stats_bench.json 2>/tmp/e1.log
DB=$(find /tmp/p_stats -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB $O/prof_stats.json
python $R/tools/gap_profile.py $DB $O/prof_gaps.json
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $CMD > /dev/null 2>/tmp/e2.log
python $R/tools/prof_summary.py $(find /tmp/p_fetch -name "*.db" | head -1) $O/prof_pmc_fetch.json
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o w -- $CMD > /dev/null 2>/tmp/e3.log
python $R/tools/prof_summary.py $(find /tmp/p_write -name "*.db" | head -1) $O/prof_pmc_write.json
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/p_sq -o q -- $CMD > /dev/null 2>/tmp/e4.log
python $R/tools/prof_summary.py $(find /tmp/p_sq -name "*.db" | head -1) $O/prof_pmc_sq.json
SCMD="python $R/bench.py --config street --steps 8 --warmup 4"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_street -o s -- $SCMD > $O/prof_street_bench.json 2>/tmp/e7.log
python $R/tools/prof_summary.py $(find /tmp/p_street -name "*.db" | head -1) $O/prof_street_stats.json
tail -c 400 $O/prof_stats_bench.json
