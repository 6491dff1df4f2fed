#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02n; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "config3_full_batch" --durations=3 ) > $O/fs1.log 2>&1
tail -n 8 $O/fs1.log
