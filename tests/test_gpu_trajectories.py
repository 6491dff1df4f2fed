"""GPU trajectory parity (BASELINE metric 2: "per-feature KL trajectories"): >= 100 optimizer steps through the beta ramp -
KL rises during pre-training and collapses towards 0 as beta is annealed - compared with the float64 oracles, for
  * the custom InfoNCE loop (reference train.py:180-289, BASELINE config 2: pendulum layout [2,1,2,1], l2 similarity, the
    reference's default networks and learning rate) - the COMPOSED step: X model + Y encoder + InfoNCE + one Keras-Adam over
    all variables + step-indexed epochs + numpy beta update + validation series;
  * the Keras path (model.fit + InfoBottleneckAnnealingCallback, train.py:138-166).

Tolerances.  The bar is BASELINE's: every per-epoch per-feature KL within 1e-3 nats (absolute) of the oracle; losses within
2e-3 relative.  An Adam trajectory on ReLU networks is a chaotic recursion: a pre-activation within round-off of 0 flips one
unit's subgradient and two float32-ACCURATE runs part by 1e-4..1e-3 from there on.  The tests therefore also run the oracle in
float32 and widen the tolerance of every series entry by 3 x |oracle64 - oracle32| at that entry: where float32 arithmetic
itself leaves the value open, the device may differ by as much - and nowhere else.  (At the seeds used here the widening is
< 2e-4 everywhere: the assertions are, in effect, the plain 1e-3.)  Final parameters: see _check_params."""
import numpy as np
import pytest
import torch

import dib_oracle as orc
from _helpers import flat_to_params, spec_kwargs

pytestmark = pytest.mark.gpu


def _check(name, got, want, want32, tol_abs, tol_rel=0.0):
    got, want, want32 = [np.asarray(a, dtype=np.float64) for a in (got, want, want32)]
    assert got.shape == want.shape, (name, got.shape, want.shape)
    tol = tol_abs + tol_rel * np.abs(want) + 3.0 * np.abs(want - want32)
    err = np.abs(got - want)
    assert (err <= tol).all(), (name, float(err.max()), float(tol.flat[np.argmax(err - tol)]), np.argwhere(err > tol)[:4])
    return float(err.max())


def _check_params(name, got, want, want32, start):
    """Final parameters of one block after n Adam steps.  Adam normalises every element's step to ~lr whatever the size of
    its gradient, so an element whose gradient is round-off-level noise moves by lr per step in a direction float32 does not
    determine: a per-element absolute bound of the order of the float32 error does not exist.  What is bounded is the error
    relative to the DISPLACEMENT the optimiser produced: rms error <= 2 % of the block's rms displacement, largest error <=
    5 % of its largest displacement (each widened by 3 x the float64-vs-float32 oracle difference, as for the series)."""
    got, want, want32, start = [np.asarray(a, dtype=np.float64) for a in (got, want, want32, start)]
    assert got.shape == want.shape, (name, got.shape, want.shape)
    disp, err, amb = want - start, got - want, want - want32
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    assert rms(err) <= 0.02 * rms(disp) + 3.0 * rms(amb) + 1e-7, (name, "rms", rms(err), rms(disp), rms(amb))
    assert np.abs(err).max() <= 0.05 * np.abs(disp).max() + 3.0 * np.abs(amb).max() + 1e-6, \
        (name, "max", float(np.abs(err).max()), float(np.abs(disp).max()), float(np.abs(amb).max()))
    return rms(err) / max(rms(disp), 1e-30)


@pytest.mark.parametrize("batch_size,rows", [(128, 1024), (256, 2048)])
def test_infonce_loop_trajectory_matches_float64_oracle(tmp_path, batch_size, rows):
    import dib_amd
    import infonce_loop_oracle as ilo
    from dib_amd import data, infonce
    from dib_amd.dense import DenseStack
    d = data.fetch_double_pendulum(data_path=str(tmp_path), pendulum_number_trajectories=6, seed=0)
    xt, yt, xv, yv = d["x_train"][:rows], d["y_train"][:rows], d["x_valid"][:300], d["y_valid"][:300]
    spec = orc.DIBSpec([2, 1, 2, 1], [128, 128], [256, 256], 64)          # train.py defaults, shared dimensionality 64
    seed, lr, n_pre, n_ann, b0, b1 = 3, 3e-4, 3, 11, 1e-3, 3.0             # 13 recorded epochs x 8 steps = 104 optimizer steps
    model = dib_amd.DistributedIBNet(**spec_kwargs(spec), noise_seed=5, init_seed=seed)
    eng = model._ensure_engine()
    yenc = DenseStack(eng, 6, [128, 128], 64, "relu", True, 5, seed=seed + 1)
    L = len(yenc.dims)
    p0 = flat_to_params(model.param_blocks(), model.get_flat_weights(), spec)
    y0 = ([yenc.kernel(l).cpu().numpy().astype(np.float64) for l in range(L)],
          [yenc.bias(l).cpu().numpy().astype(np.float64) for l in range(L)])
    got = infonce.fit_infonce(model, xt, yt, xv, yv, batch_size=batch_size, number_pretraining_epochs=n_pre,
                              number_annealing_epochs=n_ann, beta_start=b0, beta_end=b1, learning_rate=lr, similarity="l2",
                              temperature=1.0, seed=seed, output_encoder=yenc)
    runs = {}
    for dt in (torch.float64, torch.float32):
        ye = ilo.YEncoder(y0[0], y0[1], "relu", True, 5, dtype=dt)
        o = ilo.InfoNCELoopOracle(spec, p0, ye, "l2", 1.0, lr, noise_seed=5, dtype=dt)
        runs[dt] = (o.fit(xt, yt, xv, yv, batch_size=batch_size, number_pretraining_epochs=n_pre, number_annealing_epochs=n_ann,
                          beta_start=b0, beta_end=b1, seed=seed), o)
    (want, o64), (want32, o32) = runs[torch.float64], runs[torch.float32]
    assert o64.t == 104 and want["kl"].shape == (13, 4)
    # the ramp crosses the collapse: total KL peaks mid-run and ends well below its peak
    assert want["kl_total"].max() > 2.0 * want["kl_total"][-1]
    assert np.array_equal(got["beta"], want["beta"]) and got["beta"].dtype == np.float32
    errs = {}
    for k in ("kl", "kl_validation"):
        errs[k] = _check(k, got[k], want[k], want32[k], tol_abs=1e-3)
    for k in ("kl_total", "kl_total_validation"):                                # the reference's own series (sum over features)
        errs[k] = _check(k, got[k], want[k], want32[k], tol_abs=2e-3)
    for k in ("loss_infonce", "loss_infonce_validation"):
        errs[k] = _check(k, got[k], want[k], want32[k], tol_abs=0.0, tol_rel=2e-3)
    # final parameters of both networks after 104 Adam steps (lr 3e-4: a parameter moves <= 0.03 in total)
    px = flat_to_params(model.param_blocks(), model.get_flat_weights(), spec)
    nx = len(px.tensors())
    perr = {}
    for i, (t, t0) in enumerate(zip(px.tensors(), p0.tensors())):
        w64, w32 = o64.vars[i].detach().numpy(), o32.vars[i].detach().double().numpy()
        perr[f"x{i}"] = _check_params(f"x param {i}", t, w64, w32, t0)
    for l in range(L):
        for j, t in enumerate((yenc.kernel(l), yenc.bias(l))):
            w64, w32 = o64.vars[nx + 2 * l + j].detach().numpy(), o32.vars[nx + 2 * l + j].detach().double().numpy()
            perr[f"y{l}{j}"] = _check_params(f"y param {l}/{j}", t.cpu().numpy(), w64, w32, y0[j][l])
    print("max series errors:", {k: f"{v:.2e}" for k, v in errs.items()},
          "worst parameter block (rms error / rms displacement):", max(perr, key=perr.get), f"{max(perr.values()):.2e}")


def test_keras_path_trajectory_160_steps_through_the_ramp():
    """model.fit over 40 epochs x 4 steps (B = 256) with the annealing callback ramping beta 1e-3 -> 3 after 5 epochs:
    every History key (loss, KL0..7, beta, accuracy, val_*) against orc.fit, final parameters included."""
    import dib_amd
    spec = orc.DIBSpec([1] * 8, [64, 64], [128, 128], 1, feature_embedding_dimension=16)
    rng = np.random.default_rng(20241008)
    n, bs, epochs, lr = 1024, 256, 40, 1e-3
    x = rng.standard_normal((n, 8)).astype(np.float32)
    w = rng.standard_normal(8)
    y = ((x[:, :4] @ w[:4] + 0.5 * x[:, 0] * x[:, 1]) > 0).astype(np.float32)   # 4 informative, 4 pure-noise features
    model = dib_amd.DistributedIBNet(**spec_kwargs(spec), noise_seed=4, shuffle_seed=6, init_seed=2)
    opt = dib_amd.optimizers.get("adam")
    opt.learning_rate = lr
    model.compile(optimizer=opt, loss=dib_amd.losses.BinaryCrossentropy(from_logits=True), metrics=["accuracy"])
    cb = dib_amd.InfoBottleneckAnnealingCallback(1e-3, 3.0, 5, 35)
    p0 = flat_to_params(model.param_blocks(), model.get_flat_weights(), spec)
    hist = model.fit(x, y, epochs=epochs, shuffle=True, batch_size=bs, callbacks=[cb], verbose=False,
                     validation_data=(x[:300], y[:300])).history
    ref, finals = {}, {}
    for dt in (np.float64, np.float32):
        p = p0.astype(dt)
        ref[dt] = orc.fit(spec, p, x, y, epochs=epochs, batch_size=bs, loss_kind="bce_logits",
                          beta_fn=lambda e: orc.beta_schedule(e, 1e-3, 3.0, 5, 35), lr=lr, shuffle=True,
                          validation_data=(x[:300], y[:300]), noise_seed=4, shuffle_seed=6, metrics=["accuracy"], dtype=dt)
        finals[dt] = p
    want, want32 = ref[np.float64], ref[np.float32]
    assert set(want) == set(hist)
    kl = np.array([want[f"KL{f}"] for f in range(8)])
    assert kl.max() > 3.0 and kl[:, -1].max() < 0.05, "the anneal must cross the collapse point"
    errs = {}
    for k in want:
        if "accuracy" in k:   # one sample either side of the 0.5 threshold is 1/n
            rows = 300 if k.startswith("val_") else n
            assert np.abs(np.array(hist[k]) - np.array(want[k])).max() <= 3.0 / rows + 1e-9, k
        elif "KL" in k:
            errs[k] = _check(k, hist[k], want[k], want32[k], tol_abs=1e-3)
        elif "beta" in k:
            assert np.allclose(hist[k], want[k], rtol=1e-6), k
        else:
            errs[k] = _check(k, hist[k], want[k], want32[k], tol_abs=0.0, tol_rel=2e-3)
    got = flat_to_params(model.param_blocks(), model.get_flat_weights(), spec)
    perr = max(_check_params(f"param {i}", a, b, c, a0) for i, (a, b, c, a0) in enumerate(
        zip(got.tensors(), finals[np.float64].tensors(), finals[np.float32].tensors(), p0.tensors())))
    print("max series errors:", {k: f"{v:.2e}" for k, v in errs.items()}, "worst parameter block (rms error / rms displacement):", perr)
