#!/bin/bash
# flash attention backward without divergent control flow (accumulators stay in AGPRs), pinned LDS prefetch
export TMPDIR=/tmp
O=gpurun_out/r02q; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_set_transformer.py -q -x ) > $O/st_tests.log 2>&1
tail -n 3 $O/st_tests.log
for bp in "4 4096" "8 1024" "32 50" "3 1000"; do set -- $bp; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 2>&1 | tail -n 1; done | tee $O/attn_bench.txt
for at in flash; do for bp in "32 50" "4 4096"; do set -- $bp; DIB_ST_ATTENTION=$at timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1; done; done | tee $O/st_bench.txt
