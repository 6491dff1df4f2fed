#!/bin/bash
# r05r: one-launch InfoNCE kernel with 512 threads (2 waves per SIMD), v_sqrt / v_rsq, 16 x 16 gradient tiles: phase timing,
# InfoNCE tests, same-box A/B of the loop step by tuning key
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05r; mkdir -p $O
DIB_LIB_PATH=exp/lib_STIMING.so timeout 120 python tools/small_phase_timing.py infonce 2>&1 | grep dib_infonce | tee $O/infonce_phase_timing.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "infonce" ) > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
timeout 300 python tools/config2_loop_ab.py 128 2>&1 | grep '^{' | tee $O/loop_ab.txt
