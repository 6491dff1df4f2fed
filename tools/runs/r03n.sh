#!/bin/bash
# r03n: InfoNCE (coefficient matrices + transposed copies), MI row kernels (dimension-major operands), DenseStack on
# dib_gemm_grouped with split weight gradients: parity, then kernel / loop times
O=gpurun_out/r03n; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_set_transformer.py -m gpu -q -x -k "infonce or pendulum or train_script or dense_stack or mi_sandwich or probe_grid or info_per_feature" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -6 $O/pytest.log
python tools/infonce_bench.py 2>/dev/null | tee $O/infonce_bench.txt
python tools/secondary_paths_bench.py infonce mi 2>$O/bench.err | tee $O/secondary.txt
