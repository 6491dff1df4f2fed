#!/bin/bash
# per-phase cycles of the final attention backward
export TMPDIR=/tmp
mkdir -p gpurun_out/r02ao
timeout 50 python tools/attn_phase_timing.py 2>&1 | tail -n 10 | tee gpurun_out/r02ao/phases.txt
