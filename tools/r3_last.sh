#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/last_smoke.log 2>&1; tail -2 $O/last_smoke.log
timeout 600 python -m pytest tests/test_field.py tests/test_pack_ops.py tests/test_ray_query.py tests/test_trainer.py -m gpu -q > $O/last_tests.log 2>&1; tail -2 $O/last_tests.log
