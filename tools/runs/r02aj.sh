#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02aj
timeout 300 python tools/st_grad_error_table.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02aj/table.txt
