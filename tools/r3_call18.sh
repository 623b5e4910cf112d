#!/bin/bash
# GPU call 18: permutohedral encoding + field tests on the GPU, timing of a permuto field query vs LoTD
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_permuto.py tests/test_abi.py tests/test_distant.py -m gpu -x -q > $O/c18_tests.log 2>&1
tail -3 $O/c18_tests.log
python tools/field_bench.py --shape permuto --rays 8192 --per-ray 38 --iters 8 > $O/c18_fb_permuto.json 2>$O/c18.err
python tools/field_bench.py --shape object --rays 8192 --per-ray 38 --iters 8 > $O/c18_fb_object.json 2>>$O/c18.err
cat $O/c18_fb_permuto.json $O/c18_fb_object.json; tail -3 $O/c18.err
