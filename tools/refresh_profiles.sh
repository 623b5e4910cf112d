#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line + rocprofv3 kernel stats + gap profile + the HBM and SQ/MFMA PMC passes of
# the same command (counters in their own runs, --kernel-trace only), then the kernel stats of the distant-model and
# street-config steps.  Small JSON summaries land in gpurun_out/ (the raw databases stay in /tmp);
# tools/make_profiles.py turns them into the tracked files under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rm -f $O/prof_field_bench.jsonl
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
timeout 900 python $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err
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
# L2-side request counters of the same command (own pass): read requests of the gathers, atomic requests of the scatter
timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/p_l2 -o c -- $CMD > /dev/null 2>/tmp/e8.log
python $R/tools/prof_summary.py $(find /tmp/p_l2 -name "*.db" | head -1) $O/prof_pmc_l2.json
# the distant-model step and the street configuration: kernel stats (steady state: 24 / 12 steps incl. warm-up)
DCMD="python $R/bench.py --distant --steps 16 --warmup 8 --no-cpu-baseline --no-variants --no-parity"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_dist -o s -- $DCMD > $O/prof_distant_bench.json 2>/tmp/e5.log
python $R/tools/prof_summary.py $(find /tmp/p_dist -name "*.db" | head -1) $O/prof_distant_stats.json
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/p_dsq -o q -- $DCMD > /dev/null 2>/tmp/e6.log
python $R/tools/prof_summary.py $(find /tmp/p_dsq -name "*.db" | head -1) $O/prof_distant_pmc_sq.json
SCMD="python $R/bench.py --config street --steps 8 --warmup 4"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_street -o s -- $SCMD > $O/prof_street_bench.json 2>/tmp/e7.log
python $R/tools/prof_summary.py $(find /tmp/p_street -name "*.db" | head -1) $O/prof_street_stats.json
[ "${NSIM_PROFILE_EXTRAS:-0}" = "1" ] || exit 0      # the round-3 A/B legs below only on request
timeout 300 python $R/tools/scatter_levels.py $O/prof_scatter_levels.json > /dev/null 2>&1
# fused 4-D gather A/B WITHOUT a profiler on either side
timeout 300 python $R/bench.py --distant --steps 16 --warmup 8 --no-cpu-baseline --no-variants --no-parity > $O/prof_distant_lmgather.json 2>/dev/null
NSIM_DISTANT_FUSED_GATHER=1 timeout 300 python $R/bench.py --distant --steps 16 --warmup 8 --no-cpu-baseline --no-variants --no-parity > $O/prof_distant_fusedgather.json 2>/dev/null
# per-iteration s_memtime timelines of the decoder kernels (a -DNSIM_KTIME build next to the product library, if present)
[ -f $R/neuralsim_amd/csrc/_probe/libnsim_hip_ktime.so ] && (cd $R && timeout 300 python tools/ktime.py > $O/prof_ktime.txt 2>/dev/null)
# flush storm A/B: direct weight-gradient flush vs replicas, same command
NSIM_GRAD_REPLICAS=1 timeout 300 $CMD > $O/prof_bench_noreplicas.json 2>/dev/null
# per-entry-point times of a with-grad query: object / street / permuto shapes
(cd $R && for sh in object street permuto; do timeout 200 python tools/field_bench.py --shape $sh $( [ $sh = street ] || echo --rays 8192 --per-ray 38 ) >> $O/prof_field_bench.jsonl 2>/dev/null; done)
tail -1 $O/prof_bench.json | cut -c1-300
