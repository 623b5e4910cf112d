#!/bin/bash
# Final round-3 validation on the GPU: full -m gpu suite, smoke, the default bench line, street kernel stats of the final tree
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/final_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1
python bench.py > $O/final_bench.json 2>$O/final_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_street -o s -- python $R/bench.py --config street --steps 8 --warmup 4 > $O/prof_street_bench.json 2>/tmp/e7.log
python $R/tools/prof_summary.py $(find /tmp/p_street -name "*.db" | head -1) $O/prof_street_stats.json
CMD="python $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o s -- $CMD > $O/prof_stats_bench.json 2>/tmp/e1.log
DB=$(find /tmp/p_stats -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB $O/prof_stats.json
python $R/tools/gap_profile.py $DB $O/prof_gaps.json
cd $R
tail -3 $O/final_tests.log; tail -2 $O/final_smoke.log
python - <<PY
import json
d=json.loads(open("$O/final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("step_ms"), json.dumps(d.get("variants")), json.dumps(d.get("parity")), json.dumps(d.get("cpu_baseline"))[:200])
PY
