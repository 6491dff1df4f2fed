#!/bin/bash
# r03k: the GPU suite under the opt-in modes (they must pass unchanged): hipGraph replay of both models' steps, the general
# GEMM path instead of the fused encoder kernels, recompute-mode attention; notebook-size step under fwd tile forcing
O=gpurun_out/r03k; mkdir -p $O
(DIB_ENABLE_GRAPHS=1 timeout 900 python -m pytest tests -m gpu -q -k "not fullsize and not config5_size and not bf16x6" > $O/pytest_graphs.log 2>&1; echo "rc=$?" >> $O/pytest_graphs.log); tail -4 $O/pytest_graphs.log
(DIB_DISABLE_FUSED=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp_and_cache.py -m gpu -q -k "not bf16x6 and not infonce" > $O/pytest_nofused.log 2>&1; echo "rc=$?" >> $O/pytest_nofused.log); tail -4 $O/pytest_nofused.log
for t in "" "DIB_FORCE_TILE0=22" "DIB_FORCE_TILE0=11" "DIB_FORCE_TILE0=21"; do echo "[$t] $(env $t python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 50 --warmup 5 2>/dev/null)"; done | tee $O/st_tile_forcing.txt
