"""Same-box A/B of the BASELINE config-2 loop step (bench.config2_infonce_loop: the custom InfoNCE loop on the pendulum layout) with
the row-tile integration kernels' cluster mode on / off (dib_set_tuning "int_cluster").  usage: python tools/config2_cluster_ab.py [batch ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dib_amd import _lib  # noqa: E402

default = _lib.get_tuning("int_cluster")
for batch in [int(a) for a in sys.argv[1:]] or [128]:
    for rep in range(3):
        for cl in (0, default):
            _lib.set_tuning("int_cluster", cl)
            r = bench.config2_infonce_loop("cuda:0", batch)
            print(json.dumps(dict(batch=batch, int_cluster=cl, ms_per_step=r["ms_per_step"],
                                  steady=r.get("ms_per_step_256_steps_per_epoch"), launches=r["library_launches_per_step"])), flush=True)
_lib.set_tuning("int_cluster", default)
