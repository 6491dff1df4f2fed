#!/bin/bash
# re-validation of the final library after the deterministic-seed fix: whole GPU suite, parity file under bf16x6, smoke()
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
( time timeout 1700 python -m pytest tests -m gpu -q -x --durations=6 ) > $O/gpu_tests.log 2>&1
tail -n 12 $O/gpu_tests.log
( DIB_GEMM_MODE=bf16x6 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x ) > $O/gpu_tests_bf16x6.log 2>&1
tail -n 4 $O/gpu_tests_bf16x6.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
