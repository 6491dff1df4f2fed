#!/bin/bash
# r03c: score-stash attention - parity (set-transformer suite), kernel-level A/B (exp/lib_ATT0.so = product,
# lib_ATT1.so = + one-tile-ahead stash prefetch, removed afterwards), config-5 bench line
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_set_transformer.py -m gpu -q -x > gpurun_out/r03c/pytest_st.log 2>&1
for rep in 1 2; do for v in ATT0 ATT1; do for s in 1 0; do echo "$v stash=$s $(DIB_LIB_PATH=exp/lib_$v.so python tools/attn_bench.py --batch 4 --particles 4096 --stash $s)"; done; done; done 2>&1 | tee gpurun_out/r03c/attn_ab.txt
python bench.py --config5-only --steps 4 > gpurun_out/r03c/config5.json 2> gpurun_out/r03c/config5.err
