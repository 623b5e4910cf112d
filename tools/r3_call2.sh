#!/bin/bash
# round-3 GPU call 2: full-size parity at configs[2..4], distant tests, bench with the new variants
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_fullsize_configs.py -x -q -m gpu > $O/c2_configs.log 2>&1
tail -5 $O/c2_configs.log
timeout 600 python -m pytest tests/test_distant.py tests/test_fullsize_parity.py -q -m gpu > $O/c2_parity.log 2>&1
tail -3 $O/c2_parity.log
timeout 900 python bench.py --no-cpu-baseline > $O/c2_bench.json 2> $O/c2_bench.err
tail -c 1500 $O/c2_bench.json; tail -5 $O/c2_bench.err
