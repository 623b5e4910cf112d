#!/bin/bash
# round 6 (session 2), GPU call 2: the embedded-position decoder on the matrix cores (k_field / k_field_bwd_j NE = 2):
# parity tests on the device, per-entry-point times of the Vehicle shape with / without the block, kernel stats
OUT=gpurun_out/r6_s2_call2
mkdir -p $OUT
python -m pytest tests/test_field.py tests/test_batched.py tests/test_abi.py -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for sh in vehicle vehicle_noembed; do
  for n in 2048 8192; do
    python tools/field_bench.py --shape $sh --rays $n --per-ray 32 --iters 8 >> $OUT/field_bench.jsonl 2>> $OUT/field_bench.err
  done
done
cat $OUT/field_bench.jsonl
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_wide -o s -- python $R/tools/field_bench.py --shape vehicle --rays 2048 --per-ray 32 --iters 8 > /dev/null 2>/tmp/e1.log
python $R/tools/prof_summary.py $(find /tmp/p_wide -name "*.db" | head -1) $R/$OUT/wide_stats.json
python - <<PY
import json
d=json.load(open("$R/$OUT/wide_stats.json"))
for k in d["kernels"][:12]: print(f"{k['name'][:80]:80s} {k['calls']:5d} {k['avg_us']:9.2f}")
PY
