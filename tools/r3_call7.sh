#!/bin/bash
# GPU call 7: activation-cost probe: product build vs -DNSIM_PROBE_CHEAP_ACT, same bench command
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
A="--steps 32 --warmup 16 --no-cpu-baseline --no-variants --no-parity"
python bench.py $A > $O/c7_product.json 2>$O/c7_product.err
python tools/act_probe.py run $A > $O/c7_cheap_act.json 2>$O/c7_cheap.err
python bench.py $A > $O/c7_product2.json 2>>$O/c7_product.err
tail -c 1500 $O/c7_cheap_act.json
