#!/bin/bash
# r03j: InfoNCE kernels after the O(B^2 D) rewrite: parity (all InfoNCE tests incl. the pendulum end-to-end runs) + kernel times
O=gpurun_out/r03j; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "infonce or pendulum or train_script" > $O/pytest_infonce.log 2>&1; echo "rc=$?" >> $O/pytest_infonce.log); tail -6 $O/pytest_infonce.log
python tools/infonce_bench.py 2>/dev/null | tee $O/infonce_bench.txt
