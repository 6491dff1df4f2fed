#!/bin/bash
# r06m: fit with ONE synchronisation per epoch (validation accumulator of its own): the fit / History / graph / DP tests, then the
# reference-default run (bench.keras_path_default_batch + tools/small_batch_bench.py)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06m; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectories.py tests/test_gpu_dp_and_cache.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "fit or hipgraph or train_script or science or ib_flag or trajectory or rccl or info_per_feature" > $O/tests.txt 2>&1; tail -n 5 $O/tests.txt | cut -c1-250
for rep in 1 2 3; do python tools/small_batch_bench.py 2>/dev/null | tail -n 1; done | tee $O/default_batch.txt
python - <<'PY' | tee -a $O/default_batch.txt
import json, bench
for rep in range(2):
    r = bench.keras_path_default_batch("cuda:0")
    print(json.dumps({k: r[k] for k in ("us_per_train_plus_validation_step", "library_launches_per_train_plus_validation_step", "validation_batches_one_by_one")}))
PY
