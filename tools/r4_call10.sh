#!/bin/bash
# round 4, GPU call 10: run-aware packed_sort (parity + multi-object step), permuto fused-step oracle test
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_pack_ops.py tests/test_compose.py tests/test_renderer.py -m gpu -x -q > $O/c10_tests.log 2>&1; tail -2 $O/c10_tests.log
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "permuto_fused" > $O/c10_permuto_fused.log 2>&1; tail -3 $O/c10_permuto_fused.log
for i in 1 2; do timeout 300 python bench.py --config multi --steps 24 --warmup 8 > $O/c10_multi_$i.json 2> $O/c10_multi_$i.err; done
cd /tmp && export TMPDIR=/tmp
for n in 8 24; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_multi_$n -o s -- python $R/bench.py --config multi --steps $n --warmup 4 > /dev/null 2>/tmp/e_$n.log
  python $R/tools/prof_summary.py $(find /tmp/p_multi_$n -name "*.db" | head -1) $O/c10_multi_${n}_stats.json
done
python $R/tools/prof_diff.py $O/c10_multi_8_stats.json 8 $O/c10_multi_24_stats.json 24 14
python - <<'PY'
import json
for i in (1,2):
    d=json.loads(open(f"/root/repo/gpurun_out/c10_multi_{i}.json").read().strip().splitlines()[-1]); print("multi", d["ms_per_step"])
for p in ("f32","fp16"):
    try: print(p, open(f"/root/repo/gpurun_out/parity_fullsize_permuto_fused_{p}.json").read().replace("\n"," "))
    except Exception as e: print(p, "ERR", e)
PY
