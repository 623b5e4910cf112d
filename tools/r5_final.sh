#!/bin/bash
# Closing GPU call of round 5 (runs ON THE GPU BOX via gpurun): the full -m gpu suite, smoke(), a short soak, the default bench line.
# Logs land in gpurun_out/closing/ and are copied to profiles/round5_closing/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/closing
mkdir -p $O
cd $R
git rev-parse HEAD > $O/commit.txt 2>/dev/null || true
timeout 1000 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; echo "suite rc=$?" >> $O/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 300 python tools/soak.py 1500 > $O/soak.json 2> $O/soak.err; echo "soak rc=$?" >> $O/soak.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
tail -3 $O/suite.log; tail -2 $O/smoke.log; tail -c 400 $O/soak.json; tail -1 $O/soak.err; cut -c1-300 $O/bench.json; tail -1 $O/bench.err
