#!/bin/bash
# round 4, closing call: the full GPU suite on the committed HEAD, twice, each in a fresh process (logs kept under gpurun_out/),
# then __graft_entry__.smoke() and the default bench run
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
git rev-parse HEAD > $O/final.head 2>/dev/null || cat .git/HEAD > $O/final.head 2>/dev/null
for T in final_a final_b; do
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite_$T.log 2>&1; echo "suite rc=$?" >> $O/suite_$T.log; tail -2 $O/suite_$T.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/final_smoke.log 2>&1; tail -1 $O/final_smoke.log
timeout 900 python bench.py > $O/final_bench.json 2> $O/final_bench.err; tail -c 300 $O/final_bench.json
