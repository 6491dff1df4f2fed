"""GPU tests of the data-parallel plumbing on RCCL (one rank: the driver's box has one GPU), of bench.py's launcher path on
the GPU, and of the engine's workspace cache / autograd-bridge guards (round-1 advisor findings)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("batch", [1024, 2048])
def test_fit_under_rccl_one_rank_equals_single_process(tmp_path, batch):
    """fit() launched by torch.distributed.run (nccl = RCCL backend, world size 1) reproduces the plain single-process run
    bit for bit (a 1-rank sum all-reduce is the identity).  batch 2048: the three-bucket async all-reduce sequence with the
    per-bucket optimizer launches, the empty-tail-batch branch and the metric all-reduce on the real communicator.  batch
    1024: the small-batch regime - row-tile kernels, one grouped weight-gradient launch, ONE gradient bucket after the
    backward (fit's dp_small_batch_rows) and the plain optimizer entry point instead of the step's tail launch."""
    a, b = str(tmp_path / "single.npz"), str(tmp_path / "rccl.npz")
    worker = os.path.join(HERE, "_dp_gpu_worker.py")
    r = subprocess.run([sys.executable, worker, a, str(batch)], capture_output=True, text=True, timeout=400, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", _free_port(), worker, b, str(batch)], capture_output=True, text=True,
                       timeout=400, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    s, d = np.load(a), np.load(b)
    assert set(s.files) == set(d.files)
    for k in s.files:
        assert np.array_equal(s[k], d[k]), k


@pytest.mark.timeout(900)
def test_set_transformer_train_step_under_rccl_one_rank_equals_single_process(tmp_path):
    """SetTransformerDIB.train_step (BASELINE config 5's data-parallel step, notebook :419-431) launched by
    torch.distributed.run on the nccl = RCCL backend with one rank, forced onto its collective branch: neighbourhood
    sharding, global-token noise keys, gradient all-reduce and statistics all-reduce on the real communicator reproduce the
    single-process run bit for bit."""
    a, b = str(tmp_path / "single.npz"), str(tmp_path / "rccl.npz")
    worker = os.path.join(HERE, "_dp_gpu_st_worker.py")
    r = subprocess.run([sys.executable, worker, a], capture_output=True, text=True, timeout=400, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", _free_port(), worker, b], capture_output=True, text=True, timeout=400,
                       env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    s, d = np.load(a), np.load(b)
    assert int(s["t"]) == 3 and int(d["t"]) == 3
    for k in s.files:
        assert np.array_equal(s[k], d[k]), (k, s[k], d[k])


@pytest.mark.timeout(900)
def test_bench_under_launcher_reports_joined_ranks():
    """bench.py under torch.distributed.run with one rank: RCCL path, `n_gpus` = ranks that joined the communicator."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                        "--warmup", "1", "--blocks", "1", "--batch", "8192", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks_joined"] == 1 and out["scaling"] == "strong"
    assert out["value"] > 0 and out["config"]["global_batch"] == 8192 and "roofline" in out


def test_workspace_cache_is_lru_and_never_evicts_a_graph_workspace():
    from dib_amd.engine import HipEngine
    eng = HipEngine([1, 1], [32, 32], [16], 1, feature_embedding_dimension=32)
    ws = {b: eng.workspace(b).data_ptr() for b in (8, 16, 24, 32)}
    eng.workspace(8)                      # touch: 8 becomes most recent, 16 is now the LRU entry
    eng.workspace(40)                     # evicts 16, not 8
    assert 16 not in eng._ws and eng.workspace(8).data_ptr() == ws[8]
    # a workspace referenced by a captured graph is pinned
    x = torch.randn(64, 2, device=eng.device)
    y = (x[:, :1] > 0).float()
    eng.enable_step_counter(0)
    graph, stage = eng.capture_step_graph(x, y, 48, "bce_logits", 1.0 / 48, 0, True)
    p48 = eng.workspace(48).data_ptr()
    for b in (50, 51, 52, 53, 54, 55):
        eng.workspace(b)
    assert 48 in eng._ws and eng.workspace(48).data_ptr() == p48
    stage.copy_(torch.arange(48, dtype=torch.int32, device=eng.device))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.grads).all()
    # evaluation helpers use their own scratch: no step workspace of that size appears
    eng.encode_feature(0, np.zeros((77, 1), dtype=np.float32))
    assert 77 not in eng._ws


@pytest.mark.parametrize("arch", ["fused_128_128_32", "general_ragged"])
def test_three_bucket_backward_hooks_equal_the_plain_backward(arch):
    """The staged encoder backward of the data-parallel protocol (dib_encoder_bank_bwd_stage 1 -> finalize part 2 -> hook ->
    stage 2 -> finalize part 3) on the fused and on the general (ragged, odd widths) path: the hooks see final
    gradient slices at the time they are called, the buckets tile the flat buffer, and the result equals the one-call backward
    bit for bit - at a batch that uses split weight-gradient slabs and at one that does not."""
    from dib_amd.engine import HipEngine
    if arch == "fused_128_128_32":
        eng = HipEngine([1] * 6, [128, 128], [64, 32], 1, feature_embedding_dimension=32, init_seed=2)
    else:
        eng = HipEngine([2, 1, 3], [24, 40], [20], 1, feature_embedding_dimension=6, init_seed=2)
    ranges = [eng.part_range(p) for p in range(4)]
    (o0, c0), (o1, c1), (o2, c2), (o3, c3) = ranges
    assert o2 == 0 and o3 == c2 and c2 + c3 == c0 and o1 == c0 and o1 + c1 == eng.n_params
    rng = np.random.default_rng(0)
    for B in (96, 4096):
        x = eng.to_device(rng.standard_normal((B, eng.sum_d)).astype(np.float32))
        y = eng.to_device((rng.random((B, 1)) > 0.5).astype(np.float32))
        eng.set_beta(0.05)
        eng.train_step(x, y, None, 0, B, 1, 3, "bce_logits")
        torch.cuda.synchronize()
        ref = eng.grads.clone()
        seen = {}
        eng.grads.zero_()
        eng.train_step(x, y, None, 0, B, 1, 3, "bce_logits", accumulate=False,
                       on_integration_grads_ready=lambda g: seen.__setitem__("integration", g.clone()),
                       on_encoder_front_grads_ready=lambda g: seen.__setitem__("front", g.clone()))
        torch.cuda.synchronize()
        assert torch.equal(eng.grads, ref), (arch, B)
        assert torch.equal(seen["integration"], ref[o1: o1 + c1]) and torch.equal(seen["front"], ref[o2: o2 + c2])


def test_released_step_graph_unpins_its_workspace():
    from dib_amd.engine import HipEngine
    eng = HipEngine([1, 1], [32, 32], [16], 1, feature_embedding_dimension=32)
    x = torch.randn(64, 2, device=eng.device)
    y = (x[:, :1] > 0).float()
    eng.enable_step_counter(0)
    before = (eng.params.clone(), eng.t_dev.clone(), eng.step_dev.clone())
    graph, stage = eng.capture_step_graph(x, y, 48, "bce_logits", 1.0 / 48, 0, True)
    # the warm-up ran the captured sequence (fused head, Adam, counter bump) eagerly and restored every piece of state
    assert torch.equal(eng.params, before[0]) and torch.equal(eng.t_dev, before[1]) and torch.equal(eng.step_dev, before[2])
    assert 48 in eng._ws_pinned
    del graph
    eng.release_step_graph(48)
    assert 48 not in eng._ws_pinned
    for b in (50, 51, 52, 53, 54):
        eng.workspace(b)
    assert 48 not in eng._ws   # evictable again


def test_autograd_bridge_refuses_clobbered_workspace():
    import dib_amd
    model = dib_amd.DistributedIBNet([1, 1, 1], [32, 32], [16], 1, feature_embedding_dimension=32)
    x = torch.randn(20, 3)
    pred, kl_loss = model.forward_autograd(x)
    model(x)  # another forward with the same batch size overwrites the stashed activations
    with pytest.raises(RuntimeError, match="overwrote the stashed activations"):
        (pred.sum() + kl_loss).backward()
    pred, kl_loss = model.forward_autograd(x)
    model(torch.randn(21, 3))  # a different batch size has its own workspace: fine
    (pred.sum() + 2.0 * kl_loss).backward()
    assert torch.isfinite(model.flat_parameters.grad).all()
