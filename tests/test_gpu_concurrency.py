"""include/dib_hip.h "Threads" (thread-safe across distinct (device, stream, workspace) triples):
two host threads x two streams x two layouts / workspaces stepping concurrently produce the bits of the serial run; "four_small":
four threads x four streams of the reference-default step, whose integration kernel runs in cluster mode (workgroups that wait for each
other inside the kernel: the header's co-residency note)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize("mode", ["two_layouts", "same_arch", "shared_layout", "four_small"])
def test_two_threads_two_streams_equal_the_serial_run(mode):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "oracle"), HERE]))
    res = subprocess.run([sys.executable, os.path.join(HERE, "_concurrency_worker.py"), mode], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and "CONCURRENCY_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
