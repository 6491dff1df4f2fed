#!/bin/bash
O=gpurun_out/r06r; mkdir -p $O
for lib in STIMING STNOFENCE STNOSLEEP; do for c in 4 8; do
  echo "== $lib int_cluster=$c"; DIB_LIB_PATH=exp/lib_$lib.so timeout 120 python tools/small_phase_timing.py 128 int_cluster=$c int_cluster_wgs=256 2>&1 | grep -v "amdgpu.ids\|encoder\|gaps\|inside"
done; done > $O/exchange_variants.txt
cat $O/exchange_variants.txt
