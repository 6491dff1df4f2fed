#!/bin/bash
# Build a library variant for a same-box A/B (CPU side: hipcc cross-compiles gfx950 without a GPU):
#   bash tools/build_variant.sh TAG [-DFLAG=VALUE ...]      ->  exp/lib_TAG.so   (git-ignored, travels with gpurun)
# then on the box:  bash tools/ab_bench.sh BASE TAG   (tools/ab_bench.sh swaps each variant in as the product library).
# Compile-time knobs that exist today (csrc/dib_fused.h, csrc/dib_attn.h, csrc/dib_api.hip): DIB_H1_MASK, DIB_DW1_B64,
# DIB_ATTN_FWD_WAVES, DIB_BK11, DIB_BK212, DIB_SPLIT_ROWS;
# diagnostic builds: DIB_FUSED_TIMING (tools/fused_phase_timing.py), DIB_ATTN_TIMING (tools/attn_phase_timing.py).
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/exp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" \
  "$R/distributed-information-bottleneck.github.io_amd/csrc/dib_api.hip" -o "$R/exp/lib_$TAG.so" 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" || true
ls -la "$R/exp/lib_$TAG.so"
