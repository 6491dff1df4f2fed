#!/bin/bash
# r05g: phase timing of the row-tile kernels (diagnostic build), config-4 weight-gradient M / pitch / split sweep
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r05g; mkdir -p $O
DIB_LIB_PATH=$R/exp/lib_STIMING.so timeout 300 python tools/small_phase_timing.py 128 > $O/phase_b128.txt 2>&1; cat $O/phase_b128.txt | tail -n 8
DIB_LIB_PATH=$R/exp/lib_STIMING.so timeout 300 python tools/small_phase_timing.py 1024 > $O/phase_b1024.txt 2>&1; cat $O/phase_b1024.txt | tail -n 8
timeout 900 python tools/wgrad_m_sweep.py > $O/wgrad_m_sweep.txt 2>&1; cat $O/wgrad_m_sweep.txt | tail -n 40
