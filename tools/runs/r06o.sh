#!/bin/bash
# r06o: EXPERIMENT - the 8 row tiles of the reference-default integration kernel on ONE XCD (shared L2) vs spread over the 8 XCDs
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06o; mkdir -p $O
cd $R
python - <<'PY' | tee $O/one_xcd_ab.txt
import json, bench
from dib_amd import _lib
for rep in range(3):
    for v in (0, 1):
        _lib.set_tuning("small_int_one_xcd", v)
        r = bench.keras_path_default_batch("cuda:0")
        print(json.dumps(dict(small_int_one_xcd=v, us_per_pair=r["us_per_train_plus_validation_step"])), flush=True)
_lib.set_tuning("small_int_one_xcd", 0)
PY
