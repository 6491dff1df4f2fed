#!/bin/bash
# round-4 GPU call H: 128 x 128 weight-gradient tile with 32-deep K-tiles at 2 / 3 / 4 workgroups per CU (half the LDS of the
# 64-deep tile; register budget via launch bounds) against HEAD (64-deep, 2 per CU).  The label "<2, 2, 2, 64>" of the per-kernel
# table is the profiling slot of the (mode 2, 128 x 128) tile whatever its K depth.
export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
REPS=2 bash tools/ab_bench.sh HEAD4 W32x2 W32x3 W32x4 2>&1 | tee $O/wgrad_bk32_occupancy_ab.txt
