#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_lidar -o s -- python $R/tools/street_lidar_bench.py > $O/c30_lidar.txt 2>/tmp/e10.log
python $R/tools/prof_summary.py $(find /tmp/p_lidar -name "*.db" | head -1) $O/prof_lidar_stats.json
cat $O/c30_lidar.txt
python - <<PY
import json
d=json.load(open("$O/prof_lidar_stats.json"))
rows=[k for k in d["kernels"] if k["calls"] % 12 == 0 and k["calls"] <= 12*40]
for k in sorted(rows, key=lambda k:-k["total_us"])[:40]: print(k["name"][:80], k["calls"]//12, round(k["total_us"]/12,1))
PY
