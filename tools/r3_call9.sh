#!/bin/bash
# GPU call 9: field micro-benchmark (street shape) + kernel split, headline re-check after the NC==1 guard removal
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-parity --no-variants > $O/c9_object.json 2>$O/c9.err
python $R/tools/field_bench.py --shape street > $O/c9_fb_street.json 2>>$O/c9.err
python $R/tools/field_bench.py --shape object --rays 8192 --per-ray 38 > $O/c9_fb_object.json 2>>$O/c9.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_fb -o s -- python $R/tools/field_bench.py --shape street > /dev/null 2>>$O/c9.err
python $R/tools/prof_summary.py $(find /tmp/p_fb -name "*.db" | head -1) $O/c9_fb_street_stats.json
cat $O/c9_fb_street.json $O/c9_fb_object.json
python - <<PY
import json
d=json.loads(open("$O/c9_object.json").read().strip().splitlines()[-1])
print("object", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
s=json.load(open("$O/c9_fb_street_stats.json"))
for k in s["kernels"][:12]: print(k["name"][:60], k["calls"], round(k["avg_us"],1))
PY
