#!/bin/bash
# round-2 GPU call B: mask-aware full-size parity, suite on the new library, A/B of the kernel changes, B=8192 knob sweep
export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
P=distributed-information-bottleneck.github.io_amd/libdib_hip.so
cp exp/lib_NEW.so $P; touch $P
( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -s --durations=10 ) > $O/fullsize.log 2>&1
( time timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py ) > $O/suite.log 2>&1
tail -n 4 $O/fullsize.log; tail -n 4 $O/suite.log
echo "== A/B B=65536"; bash tools/ab_bench.sh BASE OLDMAP NEW
echo "== A/B B=8192";  BATCH=8192 TAG=b8192 bash tools/ab_bench.sh BASE OLDMAP NEW
echo "== knobs at B=8192 (NEW)"
for kv in "DIB_FORCE_TILE0=22" "DIB_FORCE_TILE0=11" "DIB_FORCE_TILE0=21" "DIB_FORCE_TILE1=11" "DIB_FORCE_TILE1=12" "DIB_L3_HALVE=0" "DIB_FORCE_TILE2=12" "DIB_FORCE_TILE2=11"; do
  echo "-- $kv"; env $kv BATCH=8192 TAG=b8192_${kv//=/} bash tools/ab_bench.sh NEW
done
cp exp/lib_NEW.so $P; touch $P
bash tools/collect_profiles.sh gpurun_out/r02b/prof > /dev/null 2>&1
ls gpurun_out/r02b
