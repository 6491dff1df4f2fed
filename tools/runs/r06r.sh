#!/bin/bash
O=gpurun_out/r06r; mkdir -p $O
for lib in STIMING STREP; do
  echo "== $lib int_cluster=4"; DIB_LIB_PATH=exp/lib_$lib.so timeout 120 python tools/small_phase_timing.py 128 int_cluster=4 2>&1 | grep -v "amdgpu.ids\|encoder\|gaps"
done > $O/phase_timing_repeat.txt
cat $O/phase_timing_repeat.txt
