#!/bin/bash
# round 6, GPU call 2: atomic conflict micro-benchmark, the scatter's sample set for the host sector model, A/B of the scatter's
# issue grouping (bench step, alternated), the new tests.  Everything lands in gpurun_out/r6_ab1/.
OUT=gpurun_out/r6_ab1
mkdir -p $OUT
./tools/atomic_conflict_bench > $OUT/atomic_conflict_bench.txt 2>&1
python tools/dump_scatter_points.py object > $OUT/dump_object.log 2>&1
python tools/dump_scatter_points.py street > $OUT/dump_street.log 2>&1
for rep in 1 2; do
  for g in 0 1; do
    NSIM_SCATTER_GROUP=$g python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-variants --no-parity > $OUT/bench_group${g}_$rep.json 2> $OUT/bench_group${g}_$rep.err
  done
done
for g in 0 1; do
  NSIM_SCATTER_GROUP=$g python bench.py --config street --steps 12 --warmup 6 > $OUT/street_group${g}.json 2> $OUT/street_group${g}.err
done
python -m pytest tests/test_convergence.py tests/test_sampling.py -q -m gpu -s -p no:cacheprovider > $OUT/new_tests.log 2>&1
echo "new tests rc=$?" >> $OUT/new_tests.log
grep -h '"value"' $OUT/bench_group*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d.get('ms_per_step_p50'), d['roofline']['avg_launch_ms'], d['roofline']['frac'])
"
tail -5 $OUT/new_tests.log
cat $OUT/atomic_conflict_bench.txt
for i in 1 2 3; do
  python -m pytest tests/test_fullsize_parity.py -q -m gpu -s -p no:cacheprovider -k "permuto_model" > $OUT/permuto_parity_$i.log 2>&1
  echo "permuto parity $i rc=$?"; grep -o '"rnd_grad_[a-z_]*": [0-9.e-]*' $OUT/permuto_parity_$i.log | tr '\n' ' '; echo
done
