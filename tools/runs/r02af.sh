#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02af
timeout 120 exp/stash_pattern 2>&1 | tee gpurun_out/r02af/stash_pattern.txt
