#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_dp_and_cache.py -q --durations=5 ) > $O/st_dp.log 2>&1
tail -n 15 $O/st_dp.log
( time DIB_GEMM_MODE=bf16x6 timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_dp_and_cache.py ) > $O/suite_bf16x6.log 2>&1
tail -n 5 $O/suite_bf16x6.log
for bp in "32 50" "2 2048" "4 4096"; do set -- $bp; DIB_ST_ATTENTION=gemm timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
