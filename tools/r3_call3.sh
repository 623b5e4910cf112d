#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_fullsize_configs.py -q -m gpu > $O/c3_configs.log 2>&1
tail -8 $O/c3_configs.log
cd /tmp && export TMPDIR=/tmp
for cfg in street multi indoor; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_$cfg -o s -- python $R/bench.py --config $cfg --steps 8 --warmup 4 > $O/c3_${cfg}_bench.json 2>/tmp/e_$cfg.log
  python $R/tools/prof_summary.py $(find /tmp/p_$cfg -name "*.db" | head -1) $O/c3_${cfg}_stats.json
done
tail -c 400 $O/c3_street_bench.json
