#!/bin/bash
# r06i: 4-row tiles of the integration network / plain MLP (v_mfma_f32_4x4x1): the parity, trajectory, DP and set-transformer test
# files; same-box A/B by tuning key on the reference-default pair and the config-2 loop
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectories.py tests/test_gpu_dp_and_cache.py tests/test_gpu_set_transformer.py tests/test_gpu_concurrency.py -m gpu -q -p no:cacheprovider -x > $O/tests.txt 2>&1; tail -n 25 $O/tests.txt | cut -c1-300
timeout 900 python tools/small4_ab.py 2>/dev/null | tee $O/small4_ab.txt
