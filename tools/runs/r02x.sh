#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02x
timeout 100 python tools/clock_rates.py 2>&1 | tail -n 3 | tee gpurun_out/r02x/clocks.txt
