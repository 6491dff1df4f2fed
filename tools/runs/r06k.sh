#!/bin/bash
# r06k: the independent full-size forward test (oracle's own Philox noise for all 134 M draws), the tuning-key test, the flat-tile
# and head-step tests at the final library
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06k; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -s -k "independent" > $O/fullsize_independent_forward.txt 2>&1; tail -n 6 $O/fullsize_independent_forward.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_building_blocks.py -m gpu -q -p no:cacheprovider -k "tuning_switchboard or building or small_batch_row_tile or step_tail" > $O/tests_b.txt 2>&1; tail -n 3 $O/tests_b.txt
