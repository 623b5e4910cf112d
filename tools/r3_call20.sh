#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "permuto" > $O/c20_tests.log 2>&1
tail -3 $O/c20_tests.log; grep -E "assert|Error" $O/c20_tests.log | head -5; cat $O/parity_fullsize_permuto_api_f32_compressed.json $O/parity_fullsize_permuto_api_fp16_compressed.json 2>/dev/null
