#!/bin/bash
# r03p: set transformer with the fused LayerNorm-backward entry and ping-pong gradient buffers: parity + step times
O=gpurun_out/r03p; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_dp_and_cache.py -m gpu -q -x -k "not bf16x6" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -4 $O/pytest.log
for a in "--batch 32 --particles 50 --steps 50 --warmup 5" "--batch 4 --particles 512 --steps 20" "--batch 4 --particles 4096 --features 16 --steps 4"; do python tools/set_transformer_bench.py $a; done 2>&1 | grep -v amdgpu.ids | tee $O/st_table.txt
