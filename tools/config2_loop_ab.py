"""Same-box A/B of the BASELINE config-2 loop step (bench.config2_infonce_loop) under the tuning keys of its small-batch
launch merges: "infonce_one_launch" (dib_infonce_small_kernel) and "mlp_row_tiles" (the output encoder on the row-tile kernels,
riding in the X model's integration grids).  usage: python tools/config2_loop_ab.py [batch ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dib_amd import _lib  # noqa: E402

for batch in [int(a) for a in sys.argv[1:]] or [128]:
    for rep in range(2):
        for one, mlp in ((1, 1), (0, 1), (1, 0), (0, 0)):
            _lib.set_tuning("infonce_one_launch", one)
            _lib.set_tuning("mlp_row_tiles", mlp)
            r = bench.config2_infonce_loop("cuda:0", batch)
            print(json.dumps(dict(batch=batch, infonce_one_launch=one, mlp_row_tiles=mlp, ms_per_step=r["ms_per_step"],
                                  launches=r["library_launches_per_step"])), flush=True)
_lib.set_tuning("infonce_one_launch", 1)
_lib.set_tuning("mlp_row_tiles", 1)
