#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -q > $O/final_tests.log 2>&1
tail -4 $O/final_tests.log; grep -E "^FAILED|^ERROR" $O/final_tests.log | head
