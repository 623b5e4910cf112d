#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_distributed.py -m gpu -q > $O/c27_tests.log 2>&1
tail -3 $O/c27_tests.log
# single-process RCCL (world size 1) through torchrun: the N > 1 code path of bench.py with the nccl backend
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline --no-variants --no-parity > $O/c27_torchrun.json 2>$O/c27_torchrun.err
tail -c 600 $O/c27_torchrun.json; tail -3 $O/c27_torchrun.err
