#!/bin/bash
# round-4 GPU call G: is the B = 8192 step slower than in round 3 (1.21-1.23 vs 1.16 ms on different boxes)?  Same-box A/B of the
# round-3 library (exp/lib_R03.so, built from commit 70bd96c; ABI check off - the config-3 entry points are unchanged) against HEAD
export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
export DIB_LIB_ABI_CHECK=0
REPS=2 BATCH=8192 TAG=b8192 bash tools/ab_bench.sh R03 HEAD4 2>&1 | tee $O/b8192_r03_vs_head_ab.txt
REPS=1 bash tools/ab_bench.sh R03 HEAD4 2>&1 | tee -a $O/b8192_r03_vs_head_ab.txt
