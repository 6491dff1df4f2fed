"""The kernel-coverage record of the GPU suite (profiles/r06_suite_kernel_coverage.txt, produced by tools/kernel_coverage.py from
a rocprofv3 --kernel-trace --marker-trace run of `pytest -m gpu`, one ROCTx range per test) against the library AS BUILT HERE:
a kernel symbol in libdib_hip.so's gfx950 code object that no GPU test launches - a new template instantiation, a dispatch
branch the tests stopped reaching (round 5's row-tile regime took the parity zoo off the large-batch fused kernels without
anyone noticing) - fails this test on the CPU."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "profiles", "r06_suite_kernel_coverage.txt")


@pytest.fixture(scope="module")
def cov():
    spec = importlib.util.spec_from_file_location("kernel_coverage", os.path.join(ROOT, "tools", "kernel_coverage.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.LIB):
        pytest.skip("libdib_hip.so not built")
    return mod, mod.read_record(RECORD), mod.library_kernels()


def test_every_kernel_symbol_of_the_library_is_launched_by_a_gpu_test(cov):
    mod, rec, kernels = cov
    assert len(kernels) >= 100
    missing = [k for k in kernels if rec.get(k, (0, 0, 0, 0))[1] == 0]
    assert not missing, ("kernels of libdib_hip.so that no GPU test launched (re-run tools/runs/r06cov.sh after adding a test):",
                         missing)


def test_every_compute_kernel_runs_under_an_oracle_comparing_test(cov):
    """Every MFMA kernel - fused encoder bank (all instantiations), grouped GEMM (all tile shapes), row-tile kernels, attention,
    InfoNCE, token chain - is launched by at least one test that compares with the float64 / NumPy oracle ('*' in the record);
    every other kernel by a '*' test or by a test that demands equality with a '*'-tested path ('=')."""
    mod, rec, kernels = cov
    compute = re.compile(r"dib_(fused_encoder_(fwd|bwd)|gemm|gemm_skinnyk|small_(encoder_fwd|encoder_bwd|integration)|attn_(fwd|bwd|small_fwd|"
                         r"small_bwd8)|infonce_(sim_mfma|grad_mfma|small)|st_chain_(fwd|bwd))_kernel")
    no_oracle = [k for k in kernels if compute.search(k) and rec.get(k, (0, 0, 0, 0))[2] == 0]
    assert not no_oracle, no_oracle
    unchecked = [k for k in kernels if rec.get(k, (0, 0, 0, 0))[2] + rec.get(k, (0, 0, 0, 0))[3] == 0]
    assert not unchecked, unchecked


def test_the_record_is_of_this_library(cov):
    """no stale entries: every KERNEL line names a symbol the library still has"""
    mod, rec, kernels = cov
    gone = sorted(set(rec) - set(kernels))
    assert not gone, gone
