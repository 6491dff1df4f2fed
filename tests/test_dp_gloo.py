"""CPU, world_size 2 over gloo: the data-parallel path of fit() - row-wise sharding of every global batch,
all-reduce(sum) of the flat gradient buffer, metric all-reduce - must reproduce the single-process run.
Uses the TEST-ONLY oracle engine (the HIP engine needs a GPU); the host code under test is the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(rank, world, port, out_dir, n_rows=80, buckets=3):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import dib_amd
    import dib_oracle as orc
    from _helpers import spec_kwargs
    from _oracle_engine import OracleEngine
    spec = orc.DIBSpec([1, 1, 1, 1], [8], [8], 1, feature_embedding_dimension=4)
    x, y = orc.boolean_circuit_truth_table([0, 1, 2, 3, [0, 2, 0], [2, 4, 3], [0, 5, 1]], 4)
    x = np.tile(x, (5, 1)).astype(np.float32)
    y = np.tile(y, 5).astype(np.float32)
    x, y = x[:n_rows], y[:n_rows]
    model = dib_amd.DistributedIBNet(**spec_kwargs(spec), noise_seed=1, shuffle_seed=2, init_seed=3)
    model._make_engine = lambda: OracleEngine(**model._spec_kwargs(), init_seed=model.init_seed)
    model.dp_buckets = buckets
    model.dp_small_batch_rows = 0     # exercise the requested bucket protocol at these toy batch sizes too
    opt = dib_amd.optimizers.get("adam")
    opt.learning_rate = 5e-3
    model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 0.5, 1, 2)
    hist = model.fit(x, y, epochs=3, batch_size=32, callbacks=[cb], verbose=False, validation_data=(x[:30], y[:30]))
    np.savez(os.path.join(out_dir, f"n{n_rows}_w{world}_r{rank}_b{buckets}.npz"), params=model._engine.get_flat_params(),
             **{k: np.array(v) for k, v in hist.history.items()})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("buckets", [1, 2, 3])
def test_two_rank_fit_equals_single_process(tmp_path, buckets):
    """buckets: the gradient all-reduce protocol of fit() - 1 = one all-reduce after the backward, 2 = integration bucket
    issued early + encoder bank, 3 = integration / encoder front layers / last encoder layer (the default)."""
    out = str(tmp_path)
    _run(0, 1, _free_port(), out)
    mp.spawn(_run, args=(2, _free_port(), out, 80, buckets), nprocs=2, join=True)
    ref = np.load(os.path.join(out, "n80_w1_r0_b3.npz"))
    r0 = np.load(os.path.join(out, f"n80_w2_r0_b{buckets}.npz"))
    r1 = np.load(os.path.join(out, f"n80_w2_r1_b{buckets}.npz"))
    assert np.allclose(r0["params"], r1["params"], rtol=0, atol=0), "ranks diverged"
    assert np.allclose(r0["params"], ref["params"], rtol=1e-9, atol=1e-12)
    for k in ref.files:
        if k == "params":
            continue
        assert np.allclose(r0[k], ref[k], rtol=1e-9, atol=1e-12), k
        assert np.allclose(r1[k], ref[k], rtol=1e-9, atol=1e-12), k


@pytest.mark.timeout(300)
@pytest.mark.parametrize("buckets", [2, 3])
def test_three_rank_fit_with_tail_batch_smaller_than_world(tmp_path, buckets):
    """n % batch_size = 1 on 3 ranks: two ranks have NO rows in the tail batch.  Every rank must still issue the same
    collectives (every gradient bucket, in the same order) - a mismatch hangs RCCL (round-1 advisor finding) - and the
    result must equal the single-process run."""
    out = str(tmp_path)
    _run(0, 1, _free_port(), out, 65)
    mp.spawn(_run, args=(3, _free_port(), out, 65, buckets), nprocs=3, join=True)
    ref = np.load(os.path.join(out, "n65_w1_r0_b3.npz"))
    rs = [np.load(os.path.join(out, f"n65_w3_r{r}_b{buckets}.npz")) for r in range(3)]
    for r in rs:
        assert np.array_equal(r["params"], rs[0]["params"]), "ranks diverged"
        for k in ref.files:
            assert np.allclose(r[k], ref[k], rtol=1e-9, atol=1e-12), k


def _run_infonce_dp(rank, world, port, out_dir):
    """Data-parallel InfoNCE protocol (dib_amd.infonce.infonce_data_parallel) with a torch-CPU autograd loss standing in
    for the device kernel: gathered loss = full-batch loss, concatenated local gradient rows = full-batch gradients."""
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dib_amd  # noqa: F401
    from dib_amd.infonce import infonce_data_parallel

    def infonce_cpu(ex, ey):  # symmetric CE over the -l2 similarity / T (reference train.py:203-214, utils.py:157-160)
        ex = ex.clone().requires_grad_(True)
        ey = ey.clone().requires_grad_(True)
        sim = -torch.sqrt(torch.clamp((ex * ex).sum(1, keepdim=True) + (ey * ey).sum(1)[None] - 2 * ex @ ey.T, min=0) + 1e-9) / 0.5
        tgt = torch.arange(ex.shape[0])
        loss = torch.nn.functional.cross_entropy(sim, tgt) + torch.nn.functional.cross_entropy(sim.T, tgt)
        gx, gy = torch.autograd.grad(loss, [ex, ey])
        return loss.detach(), gx, gy

    g = torch.Generator().manual_seed(0)
    ex_all, ey_all = torch.randn(12, 5, generator=g, dtype=torch.float64), torch.randn(12, 5, generator=g, dtype=torch.float64)
    b = 12 // world
    sl = slice(rank * b, (rank + 1) * b)
    loss, gx, gy = infonce_data_parallel(ex_all[sl], ey_all[sl], infonce_cpu, dist)
    full_loss, fgx, fgy = infonce_cpu(ex_all, ey_all)
    assert torch.allclose(loss, full_loss, rtol=0, atol=1e-14)
    assert torch.allclose(gx, fgx[sl], rtol=0, atol=1e-14) and torch.allclose(gy, fgy[sl], rtol=0, atol=1e-14)
    # without a process group the helper is the identity wrapper
    l1, a1, b1 = infonce_data_parallel(ex_all, ey_all, infonce_cpu, None)
    assert torch.equal(l1, full_loss) and torch.equal(a1, fgx) and torch.equal(b1, fgy)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_infonce_data_parallel_gather_protocol(tmp_path):
    mp.spawn(_run_infonce_dp, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_run_infonce_dp, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)


def _infonce_loop_case():
    """pendulum-shaped toy of BASELINE config 2: features [2, 1, 2, 1] -> 6-d shared space, Y encoder 6 -> [8] -> 6."""
    import dib_oracle as orc
    spec = orc.DIBSpec([2, 1, 2, 1], [8], [8], 6, feature_embedding_dimension=3, number_positional_encoding_frequencies=3)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((64, 6)).astype(np.float32)
    y = (x + 0.3 * rng.standard_normal((64, 6))).astype(np.float32)
    xv, yv = x[:24] + 0.1, y[:24] - 0.1
    y_in = 6 * 3                                                 # positional encoding: x and sin(f x) for 2 frequencies
    yk = [rng.uniform(-0.4, 0.4, (y_in, 8)), rng.uniform(-0.4, 0.4, (8, 6))]
    yb = [0.1 * rng.standard_normal(8), 0.1 * rng.standard_normal(6)]
    kw = dict(batch_size=16, number_pretraining_epochs=2, number_annealing_epochs=3, beta_start=1e-3, beta_end=2.0)
    return spec, orc.glorot_uniform_init(spec, 5), yk, yb, (x, y, xv, yv), kw


def _run_infonce_loop(rank, world, port, out_dir):
    """dib_amd.infonce.fit_infonce - the product's loop, data-parallel branch included - over the float64 checker engines
    of tests/_oracle_infonce_engine.py."""
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import dib_amd
    from dib_amd.infonce import fit_infonce
    from _helpers import spec_kwargs
    from _oracle_infonce_engine import CheckerXEngine, CheckerYEncoder
    spec, xp, yk, yb, data, kw = _infonce_loop_case()
    model = dib_amd.DistributedIBNet(**spec_kwargs(spec), noise_seed=21)
    model._make_engine = lambda: CheckerXEngine(spec, xp)
    yenc = CheckerYEncoder(yk, yb, "relu", True, 3)
    out = fit_infonce(model, *data, learning_rate=2e-3, shared_dimensionality=6, similarity="l2", temperature=0.7, seed=4,
                      output_encoder=yenc, **kw)
    # the custom loop keeps its KL sums in accumulators of its own - the model's History accumulator (what a later
    # model.fit reads its first epoch from) is never touched, although the loop ends one step short of the last boundary
    assert float(model._engine.metrics_acc.abs().max()) == 0.0
    np.savez(os.path.join(out_dir, f"loop_w{world}_r{rank}.npz"), x_params=model._engine.flat_params(),
             y_params=yenc.flat_params(), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_infonce_loop_data_parallel_equals_float64_loop_oracle(tmp_path):
    """The COMPOSED custom loop (reference train.py:180-289) through the product's host code at world sizes 1, 2 and 4 against
    the independent single-process float64 oracle of the loop: every series it returns and the final parameters of both
    networks.  (The gather-protocol test above checks the sharded loss / gradient rows against a full-batch autograd; this
    one checks what fit_infonce builds around it - shard selection, KL scaling by the GLOBAL batch, the two parameter
    all-reduces, one Adam step per network per training step, validation without gradients, the beta hand-over.)"""
    sys.path[:0] = [p for p in (os.path.join(os.path.dirname(HERE), "oracle"), HERE) if p not in sys.path]
    from infonce_loop_oracle import InfoNCELoopOracle, YEncoder
    spec, xp, yk, yb, data, kw = _infonce_loop_case()
    oracle = InfoNCELoopOracle(spec, xp, YEncoder(yk, yb, "relu", True, 3), "l2", 0.7, 2e-3, noise_seed=21)
    want = oracle.fit(*data, seed=4, **kw)
    want_x = np.concatenate([t.detach().numpy().reshape(-1) for t in oracle.model.tensors()])
    want_y = np.concatenate([t.detach().numpy().reshape(-1) for t in oracle.yenc.tensors()])
    out = str(tmp_path)
    _run_infonce_loop(0, 1, _free_port(), out)
    mp.spawn(_run_infonce_loop, args=(2, _free_port(), out), nprocs=2, join=True)
    mp.spawn(_run_infonce_loop, args=(4, _free_port(), out), nprocs=4, join=True)
    assert want["kl"].shape == (4, 4) and want["beta"][-1] > want["beta"][0]
    for world in (1, 2, 4):
        runs = [np.load(os.path.join(out, f"loop_w{world}_r{r}.npz")) for r in range(world)]
        for r in runs:
            assert np.array_equal(r["x_params"], runs[0]["x_params"]) and np.array_equal(r["y_params"], runs[0]["y_params"]), \
                "ranks diverged"
            for k in ("beta", "kl", "loss_infonce", "kl_validation", "loss_infonce_validation", "kl_total", "kl_total_validation"):
                assert np.allclose(r[k], want[k], rtol=1e-9, atol=1e-11), (world, k, r[k], want[k])
            assert np.allclose(r["x_params"], want_x, rtol=0, atol=1e-9) and np.allclose(r["y_params"], want_y, rtol=0, atol=1e-9)


def _run_set_transformer_dp(rank, world, port, out_dir, batch):
    """SetTransformerDIB.train_step (notebook `train_step`, DP over neighbourhoods - BASELINE config 5) on the CPU checker
    backend: product host code, oracle arithmetic."""
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import set_transformer_oracle as sto
    from _oracle_set_transformer import make_model
    spec = sto.SetTransformerSpec(particle_encoder_arch_spec=[16], bottleneck_dimension=8, key_dim=4, number_heads_per_mha=2,
                                  number_attention_blocks=2, ff_arch_per_block=[12, 8], final_processing_arch=[10])
    m = make_model(spec, init_seed=4, noise_seed=9)
    rng = np.random.default_rng(0)
    P = 7
    feats = rng.standard_normal((batch, P, 12)).astype(np.float32)
    y = (rng.random((batch, 1)) > 0.5).astype(np.float32)
    series = []
    for step in range(4):
        m.lr_dev.fill_(m.learning_rate_schedule(step + 1, 1e-2, 10))
        m.beta_dev.fill_(m.beta_schedule(step, 1e-3, 1e-1, 4))
        bce = m.train_step(feats, y)
        series.append([float(bce.item()), float(m.last["kl"].item())])
    val = m.train_step(feats[: max(1, batch - 1)], y[: max(1, batch - 1)], training=False)   # validation pass: no update
    series.append([float(val.item()), float(m.last["kl"].item())])
    np.savez(os.path.join(out_dir, f"st_b{batch}_w{world}_r{rank}.npz"), params=m.params.numpy(), series=np.array(series),
             t=int(m.t_dev.item()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,batch", [(2, 6), (3, 2)])   # (3, 2): one rank has NO neighbourhood - same collectives on all
def test_set_transformer_train_step_data_parallel_equals_single_process(tmp_path, world, batch):
    out = str(tmp_path)
    _run_set_transformer_dp(0, 1, _free_port(), out, batch)
    mp.spawn(_run_set_transformer_dp, args=(world, _free_port(), out, batch), nprocs=world, join=True)
    ref = np.load(os.path.join(out, f"st_b{batch}_w1_r0.npz"))
    rs = [np.load(os.path.join(out, f"st_b{batch}_w{world}_r{r}.npz")) for r in range(world)]
    assert int(ref["t"]) == 4
    for r in rs:
        assert np.array_equal(r["params"], rs[0]["params"]), "ranks diverged"
        assert int(r["t"]) == 4
        assert np.allclose(r["params"], ref["params"], rtol=1e-9, atol=1e-12)
        assert np.allclose(r["series"], ref["series"], rtol=1e-9, atol=1e-12), (r["series"], ref["series"])
