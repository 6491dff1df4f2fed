#!/bin/bash
# round-4 GPU call N: 64-deep K-tiles for the 64 x 64 forward (and dgrad) tile in the latency-bound small-batch regime:
# default-batch Keras path, config-2 loop, notebook-size set transformer; HEAD5 = the committed library
export TMPDIR=/tmp
O=gpurun_out/r04n; mkdir -p $O
for rep in 1 2; do
for v in HEAD5 BK11x64 BK11Dx64; do
  echo -n "$v: "
  DIB_LIB_PATH=exp/lib_$v.so python tools/small_batch_bench.py 2>/dev/null | tail -n 1 | sed 's/.*-> //; s/ per train.*//' | tr '\n' ' '
  DIB_LIB_PATH=exp/lib_$v.so python tools/config2_loop_trace.py 2048 2>/dev/null | tail -n 1 | python -c "import json,sys; print('c2@2048', json.loads(sys.stdin.read())['ms_per_step'], end=' ')"
  DIB_LIB_PATH=exp/lib_$v.so python tools/config2_loop_trace.py 128 2>/dev/null | tail -n 1 | python -c "import json,sys; print('c2@128', json.loads(sys.stdin.read())['ms_per_step'], end=' ')"
  DIB_LIB_PATH=exp/lib_$v.so python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; print('st32x50', json.loads(sys.stdin.read())['ms_per_step'])"
done; done 2>&1 | tee $O/bk11_small_batch_ab.txt
