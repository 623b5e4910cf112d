#!/bin/bash
# full GPU suite on the tree as shipped, in a fresh process; $1 = log tag.  Then the permuto fused-vs-autograd test 5x
# (it compared two independently pre-trained models; fresh processes so allocator / atomics order differ).
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-a}
git rev-parse HEAD > $O/suite_${TAG}.head 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite_${TAG}.log 2>&1; echo "suite rc=$?" >> $O/suite_${TAG}.log; tail -3 $O/suite_${TAG}.log
if [ "$2" = "repeat" ]; then
  for i in 1 2 3 4 5; do
    timeout 300 python -m pytest tests/test_permuto.py tests/test_trainer.py -m gpu -q -k "fused_step_equals" > $O/permuto_rep_$i.log 2>&1; tail -1 $O/permuto_rep_$i.log
  done
fi
