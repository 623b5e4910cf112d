#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/ktime.py > $O/c12_ktime.txt 2>$O/c12.err
cat $O/c12_ktime.txt
