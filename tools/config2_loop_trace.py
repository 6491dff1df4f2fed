"""The BASELINE config-2 loop step (bench.py's config2_infonce_loop) on its own, for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats -d OUT -o kt -- python tools/config2_loop_trace.py [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.config2_infonce_loop("cuda:0", int(sys.argv[1]) if len(sys.argv) > 1 else 2048)))
