#!/bin/bash
# GPU call 10: fixed cost (prologue + weight-gradient flush) vs per-point cost of the decoder launches: object shape at 4 sizes
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
for rays in 1024 4096 8192 16384 32768; do
  python tools/field_bench.py --shape object --rays $rays --per-ray 38 --iters 8 >> $O/c10_sizes.jsonl 2>>$O/c10.err
done
cat $O/c10_sizes.jsonl
