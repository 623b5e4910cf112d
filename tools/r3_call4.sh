#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_fullsize_configs.py tests/test_fullsize_parity.py tests/test_field.py tests/test_ray_query.py -q -m gpu > $O/c4_tests.log 2>&1
tail -6 $O/c4_tests.log
B="--no-cpu-baseline --no-variants --no-parity"
timeout 300 python bench.py --steps 64 --warmup 16 $B > $O/c4_bench_split.json 2> $O/c4_bench_split.err
NSIM_SAMPLING_PRECISION=fp16 timeout 300 python bench.py --steps 64 --warmup 16 $B > $O/c4_bench_fp16.json 2> /dev/null
NSIM_SAMPLING_PRECISION=f32 timeout 300 python bench.py --steps 64 --warmup 16 $B > $O/c4_bench_f32.json 2> /dev/null
timeout 300 python bench.py --config street --steps 12 --warmup 6 > $O/c4_street_mask.json 2>/dev/null
NSIM_DISTANT_BWD_THRE=0 timeout 300 python bench.py --config street --steps 12 --warmup 6 > $O/c4_street_nomask.json 2>/dev/null
for f in c4_bench_split c4_bench_fp16 c4_bench_f32 c4_street_mask c4_street_nomask; do
python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], {k:(v["avg_ms"],v["calls"]) for k,v in d["kernels"].items()})
PY
done
