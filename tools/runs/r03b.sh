#!/bin/bash
# r03b: set-transformer GPU tests (incl. the 4096-particle parity test and the 1-rank RCCL train_step test), InfoNCE / MI
# tests at working sizes, BASELINE config 5 bench line and its four rocprofv3 passes (recompute-mode attention backward)
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_set_transformer.py tests/test_gpu_dp_and_cache.py -m gpu -q -x -s -k "not bf16x6" > gpurun_out/r03b/pytest_st.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "infonce or mi_sandwich" > gpurun_out/r03b/pytest_infonce.log 2>&1
python bench.py --config5-only --steps 4 > gpurun_out/r03b/config5.json 2> gpurun_out/r03b/config5.err
CONFIG5=1 bash tools/collect_profiles.sh gpurun_out/r03b/p > gpurun_out/r03b/collect.log 2>&1
# CPU side afterwards: python tools/summarize_profiles.py gpurun_out/r03b/p r03b_config5 @config5
