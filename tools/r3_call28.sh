#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_multi -o s -- python $R/bench.py --config multi --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants > $O/prof_multi_bench.json 2>/tmp/e9.log
python $R/tools/prof_summary.py $(find /tmp/p_multi -name "*.db" | head -1) $O/prof_multi_stats.json
python - <<PY
import json
d=json.load(open("$O/prof_multi_stats.json"))
tot=sum(k["total_us"] for k in d["kernels"])
print("total us", tot)
for k in d["kernels"][:45]: print(k["name"][:70], k["calls"], round(k["total_us"]/12,1), round(k["avg_us"],1))
PY
