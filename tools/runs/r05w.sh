#!/bin/bash
# r05w: row-tile kernels up to 2048 rows (the chaos notebook's batch) instead of 1024: same-box A/B of the InfoNCE loop step and of
# the Keras-path step at B = 2048 (variant library -DDIB_SMALL_MAX_BATCH=2048), equivalence test under the variant
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05w; mkdir -p $O
ab() {
  timeout 200 python -c "
import json, bench, numpy as np, torch, time
print('infonce loop ms B=2048', bench.config2_infonce_loop('cuda:0', 2048)['ms_per_step'], bench.config2_infonce_loop('cuda:0', 2048)['ms_per_step'], 'B=1536', bench.config2_infonce_loop('cuda:0', 1536)['ms_per_step'])
from dib_amd.engine import HipEngine
for F, B in ((10, 2048), (64, 2048)):
    eng = HipEngine([1] * F, [128, 128], [256, 256], 1, feature_embedding_dimension=32)
    rng = np.random.default_rng(0)
    x = eng.to_device(rng.standard_normal((B, F)).astype(np.float32)); y = eng.to_device((rng.random((B, 1)) > 0.5).astype(np.float32))
    for it in range(10): eng.train_step(x, y, None, 0, B, 1, it, 'bce_logits', optimizer=('adam', 0.9, 0.999, 1e-7))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(200): eng.train_step(x, y, None, 0, B, 1, it, 'bce_logits', optimizer=('adam', 0.9, 0.999, 1e-7))
    torch.cuda.synchronize(); print('keras-path train step us  F', F, 'B', B, round((time.perf_counter() - t0) / 200 * 1e6, 1))
" 2>&1 | grep -v amdgpu.ids
}
for rep in 1 2; do
  echo "== row tiles up to 1024 rows (HEAD)"; ab
  echo "== row tiles up to 2048 rows"; DIB_LIB_PATH=exp/lib_SB2048.so ab
done 2>&1 | tee $O/ab.txt
