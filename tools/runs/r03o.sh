#!/bin/bash
# r03o: InfoNCE gradient kernel with 4 rows per workgroup: parity + times
O=gpurun_out/r03o; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "infonce or pendulum or train_script" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log); tail -4 $O/pytest.log
python tools/infonce_bench.py --batches 128 2048 --dims 64 2>/dev/null | tee $O/infonce_bench.txt
python tools/secondary_paths_bench.py infonce 2>/dev/null | tee $O/secondary.txt
