#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line + rocprofv3 kernel stats + the two HBM PMC passes of the same command.
# Small JSON summaries land in gpurun_out/ (the raw databases stay in /tmp); tools/make_profiles.py turns them into
# the tracked files under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline"
timeout 600 python $R/bench.py > $O/prof_bench.json 2> $O/prof_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o s -- $CMD > $O/prof_stats_bench.json 2>/tmp/e1.log
python $R/tools/prof_summary.py $(find /tmp/p_stats -name "*.db" | head -1) $O/prof_stats.json
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o f -- $CMD > /dev/null 2>/tmp/e2.log
python $R/tools/prof_summary.py $(find /tmp/p_fetch -name "*.db" | head -1) $O/prof_pmc_fetch.json
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o w -- $CMD > /dev/null 2>/tmp/e3.log
python $R/tools/prof_summary.py $(find /tmp/p_write -name "*.db" | head -1) $O/prof_pmc_write.json
tail -1 $O/prof_bench.json | cut -c1-300
