#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > $O/c5_tests.log 2>&1
tail -8 $O/c5_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/c5_smoke.log 2>&1; tail -3 $O/c5_smoke.log
( time timeout 1500 python bench.py ) > $O/c5_bench.json 2> $O/c5_bench.err
tail -c 2500 $O/c5_bench.json; tail -5 $O/c5_bench.err
