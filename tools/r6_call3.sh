#!/bin/bash
OUT=gpurun_out/r6_call3
mkdir -p $OUT
python -m pytest tests/test_convergence.py -q -m gpu -s -p no:cacheprovider > $OUT/convergence.log 2>&1
echo "convergence rc=$?"; grep "matched psnr" $OUT/convergence.log; tail -3 $OUT/convergence.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; python -c "
import json,sys
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('ms_per_step_p50')); print(json.dumps(d.get('variants'),indent=0))
"
