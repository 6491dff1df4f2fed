#!/bin/bash
# r06n: same-box A/B of fit's epoch boundary: two synchronisations per epoch (round 5) vs one
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06n; mkdir -p $O
cd $R
python tools/fit_sync_ab.py 600 2>/dev/null | tee $O/fit_sync_ab.txt
