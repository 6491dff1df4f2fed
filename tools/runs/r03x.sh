#!/bin/bash
# r03x: runtime cache-policy rule of the GEMM (DIB_GEMM_STREAM_ROWS: 8192 default, 1073741824 = off) and non-temporal loads of
# the fused backward's read-once tiles (exp/lib_R1.so) - same-box A/B
O=gpurun_out/r03x; mkdir -p $O
for rep in 1 2; do
  DIB_GEMM_STREAM_ROWS=1073741824 TAG=off bash tools/ab_bench.sh R0; bash tools/ab_bench.sh R0 R1
done 2>&1 | tee $O/ab.log
for rep in 1 2; do DIB_GEMM_STREAM_ROWS=1073741824 BATCH=8192 TAG=off_b8192 bash tools/ab_bench.sh R0; BATCH=8192 TAG=b8192 bash tools/ab_bench.sh R0; done 2>&1 | tee -a $O/ab.log
for s in 1073741824 8192; do for rep in 1 2; do echo "STREAM_ROWS=$s 32x50 $(DIB_GEMM_STREAM_ROWS=$s python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 50 --warmup 5 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") config5 $(DIB_GEMM_STREAM_ROWS=$s python bench.py --config5-only --steps 3 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")"; done; done | tee $O/st.txt
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm or forward_backward_parity or split_batch or dense_stack" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -2 $O/pytest.log
