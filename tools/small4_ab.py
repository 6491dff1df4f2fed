"""Same-box A/B of the 4-row tiles of the integration network / plain MLP (dib_set_tuning "small_rows4", "small4_max_rows"):
the reference-default Keras path (bench.keras_path_default_batch: F = 10, B = 128) and the BASELINE config-2 loop step
(bench.config2_infonce_loop) at B = 128 and 256.  usage: python tools/small4_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dib_amd import _lib  # noqa: E402

for rep in range(2):
    for rows4 in (0, 1):
        _lib.set_tuning("small_rows4", rows4)
        r = bench.keras_path_default_batch("cuda:0")
        print(json.dumps(dict(what="keras default pair", small_rows4=rows4, us_per_pair=r["us_per_train_plus_validation_step"],
                              one_by_one=r["validation_batches_one_by_one"]["us_per_train_plus_validation_step"])), flush=True)
        for batch in (128, 256, 512):
            c = bench.config2_infonce_loop("cuda:0", batch)
            print(json.dumps(dict(what="config-2 loop", batch=batch, small_rows4=rows4, ms_per_step=c["ms_per_step"],
                                  launches=c["library_launches_per_step"])), flush=True)
_lib.set_tuning("small_rows4", 1)
for mx in (128, 512, 1024):
    _lib.set_tuning("small4_max_rows", mx)
    for batch in (256, 512, 1024):
        c = bench.config2_infonce_loop("cuda:0", batch)
        print(json.dumps(dict(what="config-2 loop", batch=batch, small4_max_rows=mx, ms_per_step=c["ms_per_step"])), flush=True)
_lib.set_tuning("small4_max_rows", 256)
