import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- kernel-coverage record (tools/kernel_coverage.py, profiles/r06_suite_kernel_coverage.txt) ----------------------------
# With DIB_COVERAGE_LOG=<file> every test is bracketed by a ROCTx range named after its node id (rocprofv3 --marker-trace
# records it next to --kernel-trace's dispatches) and by a line of host clock readings in <file> (fallback when the marker
# trace is unavailable).  The device is drained before the range closes so that a test's kernels start inside its range.
_COV_LOG = os.environ.get("DIB_COVERAGE_LOG")
_roctx = None


def _cov_clocks():
    import time
    return [time.clock_gettime_ns(c) for c in (time.CLOCK_MONOTONIC, time.CLOCK_BOOTTIME, time.CLOCK_REALTIME)]


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    global _roctx
    if not _COV_LOG:
        yield
        return
    import ctypes
    if _roctx is None:
        try:
            _roctx = ctypes.CDLL("librocprofiler-sdk-roctx.so")
        except OSError:
            _roctx = False
    t0 = _cov_clocks()
    if _roctx:
        _roctx.roctxRangePushA(("dibtest::" + item.nodeid).encode())
    yield
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except Exception:
        pass
    if _roctx:
        _roctx.roctxRangePop()
    t1 = _cov_clocks()
    with open(_COV_LOG, "a") as f:
        f.write("\t".join([item.nodeid] + [str(v) for v in t0 + t1]) + "\n")
