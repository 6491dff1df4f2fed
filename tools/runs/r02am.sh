#!/bin/bash
# final attention numbers (forward at 2 waves/SIMD) + set-transformer table + the attention-facing GPU tests
export TMPDIR=/tmp
O=gpurun_out/r02am; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_set_transformer.py -q -x ) 2>&1 | tail -n 2 | tee $O/tests.txt
for bp in "4 4096" "8 1024" "32 50"; do set -- $bp; timeout 120 python tools/attn_bench.py --batch $1 --particles $2 --reps 10 2>&1 | tail -n 1; done | tee $O/attn_bench.txt
for bp in "32 50" "4 512" "2 2048" "4 4096"; do set -- $bp; timeout 300 python tools/set_transformer_bench.py --batch $1 --particles $2 --steps 5 2>&1 | tail -n 1 | cut -c1-230; done | tee $O/st_bench.txt
