#!/bin/bash
# GPU call 17: full -m gpu suite (NaN-poisoned empties), smoke, full bench line
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/c17_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/c17_smoke.log 2>&1
python bench.py > $O/c17_bench.json 2>$O/c17_bench.err
tail -3 $O/c17_tests.log; tail -2 $O/c17_smoke.log
python - <<PY
import json
d=json.loads(open("$O/c17_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d.get("variants")), json.dumps(d.get("parity")), json.dumps(d.get("cpu_baseline"))[:300], json.dumps(d["roofline"])[:300])
PY
