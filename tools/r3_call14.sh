#!/bin/bash
# GPU call 14: forward-decoder grid (rounds of workgroups vs persistent), replica count
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/c14_fb.jsonl
for g in 1024 512 256; do
  echo "fwd_grid $g" >> $O/c14_fb.jsonl
  NSIM_FWD_GRID=$g python tools/field_bench.py --shape object --rays 8192 --per-ray 38 --iters 12 >> $O/c14_fb.jsonl 2>>$O/c14.err
  NSIM_FWD_GRID=$g python tools/field_bench.py --shape street >> $O/c14_fb.jsonl 2>>$O/c14.err
done
for r in 8 32 64; do
  echo "replicas $r" >> $O/c14_fb.jsonl
  NSIM_GRAD_REPLICAS=$r python tools/field_bench.py --shape object --rays 8192 --per-ray 38 --iters 12 >> $O/c14_fb.jsonl 2>>$O/c14.err
done
cut -c1-330 $O/c14_fb.jsonl
