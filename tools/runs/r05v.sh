#!/bin/bash
# r05v: row-tile kernels with two weight batches in flight behind the one being multiplied (DIB_SMALL_DEPTH 3; NT = 4 forward +
# every backward instance): equivalence / parity tests on the new default, same-box A/B against the depth-2 build
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05v; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_set_transformer.py -q -x -m gpu -k "small_batch or dense or companion or step_tail or workspace_needs or token_chain or forward_backward_parity or north_star" ) > $O/tests.txt 2>&1; tail -n 5 $O/tests.txt
ab() {
  timeout 200 python -c "
import json, bench
k = bench.keras_path_default_batch('cuda:0')
print('keras pair us', k['us_per_train_plus_validation_step'], 'one-by-one', k['validation_batches_one_by_one']['us_per_train_plus_validation_step'])
print('infonce loop ms', bench.config2_infonce_loop('cuda:0', 128)['ms_per_step'], bench.config2_infonce_loop('cuda:0', 128)['ms_per_step'], 'B=1024', bench.config2_infonce_loop('cuda:0', 1024)['ms_per_step'])" 2>&1 | grep -v amdgpu.ids
  timeout 100 python tools/set_transformer_bench.py --batch 32 --particles 50 --steps 80 --warmup 8 2>/dev/null | tail -n 1 | python -c "import json,sys; print('set transformer ms', json.loads(sys.stdin.read())['ms_per_step'])"
}
for rep in 1 2; do
  echo "== depth 3 (HEAD)"; ab
  echo "== depth 2"; DIB_LIB_PATH=exp/lib_D2.so ab
done 2>&1 | tee $O/ab.txt
