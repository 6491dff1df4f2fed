#!/bin/bash
# r03t: 128x64 wgrad tile (last encoder layer) with 64-deep K-tiles (exp/lib_K64.so) vs 32 (exp/lib_K32.so = product),
# with and without the halved-splits rule
O=gpurun_out/r03t; mkdir -p $O
REPS=2 STEPS=20 bash tools/ab_bench.sh K32 K64 2>&1 | tee $O/ab.log
DIB_L3_HALVE=0 TAG=nohalve REPS=2 STEPS=20 bash tools/ab_bench.sh K32 K64 2>&1 | tee -a $O/ab.log
BATCH=8192 TAG=b8192 REPS=2 bash tools/ab_bench.sh K32 K64 2>&1 | tee -a $O/ab.log
