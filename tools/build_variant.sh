#!/bin/bash
# Build a library variant for a same-box A/B (CPU side: hipcc cross-compiles gfx950 without a GPU):
#   bash tools/build_variant.sh TAG [-DFLAG=VALUE ...]      ->  exp/lib_TAG.so   (git-ignored, travels with gpurun)
# then on the box:  bash tools/ab_bench.sh BASE TAG   (tools/ab_bench.sh swaps each variant in as the product library).
# Compile-time flags that exist today are the three diagnostic builds: DIB_FUSED_TIMING (tools/fused_phase_timing.py),
# DIB_ATTN_TIMING (tools/attn_phase_timing.py), DIB_SMALL_TIMING (tools/small_phase_timing.py); an A/B of a kernel change adds its
# own #ifdef for the experiment and removes it with the decision (the record stays in profiles/*_ab.txt).  Run-time choices are
# dib_set_tuning keys (include/dib_hip.h; `python bench.py --tuning key=value`) - the library reads no environment variable.
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/exp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" \
  "$R/distributed-information-bottleneck.github.io_amd/csrc/dib_api.hip" -o "$R/exp/lib_$TAG.so" 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" || true
ls -la "$R/exp/lib_$TAG.so"
