#!/bin/bash
# r05u: HEAD with the 8-wave attention backward as default and the row-tile loader issuing 4 loads per trip: parity +
# set-transformer test files, the three small-batch bench functions, notebook-size set-transformer step
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05u; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_set_transformer.py -q -x -m gpu ) > $O/tests.txt 2>&1; tail -n 6 $O/tests.txt
for rep in 1 2; do timeout 100 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 80 --warmup 8 2>/dev/null | tail -n 1; done | tee $O/st.txt
timeout 100 python tools/set_transformer_bench.py --batch 8 --particles 200 --steps 40 --warmup 8 2>/dev/null | tail -n 1 | tee -a $O/st.txt
timeout 300 python -c "
import json, bench
print(json.dumps(bench.keras_path_default_batch('cuda:0')))
print(json.dumps(bench.config2_infonce_loop('cuda:0', 128)))
print(json.dumps(bench.config2_infonce_loop('cuda:0', 128)))
print(json.dumps(bench.config2_infonce_loop('cuda:0', 1024)))" 2>&1 | grep '^{' | tee $O/small.txt
