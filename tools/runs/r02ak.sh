#!/bin/bash
# validation of the final attention backward (exponentials under the dV MFMAs): set-transformer + parity + DP/cache test files
export TMPDIR=/tmp
mkdir -p gpurun_out/r02ak
( timeout 900 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_parity.py tests/test_gpu_dp_and_cache.py -q -x ) 2>&1 | tail -n 4 | tee gpurun_out/r02ak/tests.txt
for bp in "4 4096" "8 1024" "32 50" "3 1000"; do set -- $bp; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 --reps 10 2>&1 | tail -n 1; done | tee gpurun_out/r02ak/attn_bench.txt
timeout 200 python tools/set_transformer_bench.py --batch 4 --particles 4096 --steps 5 2>&1 | tail -n 1 | cut -c1-230 | tee gpurun_out/r02ak/st.txt
