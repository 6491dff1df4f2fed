"""bench.py --gpus N must really start N ranks (round-1 verdict: it silently ran one).  CPU dry run over gloo: the same
re-exec-under-torch.distributed.run path the GPU run takes, stopping after the rendezvous + one all-reduce."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_bench_gpus_flag_spawns_that_many_ranks():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-backend", "gloo"],
                         capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"dry_run": True, "n_gpus": 2, "ranks_joined": 2}


@pytest.mark.timeout(300)
def test_bench_refuses_world_size_mismatch():
    """Launched by a launcher with a different rank count than --gpus: refuse instead of printing a wrong n_gpus."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run-backend", "gloo"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr


def test_flop_model_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.gemm_flops_per_sample(64) == 13141504   # SURVEY 8(a) totals, config 3
    assert bench.gemm_flops_per_sample(50) == 10353152   # config 4
    assert sum(bench.flops_by_kernel().values()) == 13141504
