"""Two more pins obtained by EXECUTING reference code (build container only; numbers stored, no source copied):
  * visualization.save_distributed_info_plane (visualization.py:83-113), lifted by AST and run with the recording matplotlib
    stand-in: the sieve / start-index rule and the arrays handed to every ax.plot call (History post-processing -> figure);
  * chaos/chaos_data.generate_data (chaos/chaos_data.py:3-55) for the three maps under a seeded global NumPy stream.
    python tests/golden/make_golden_misc.py  ->  tests/golden/misc.npz"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from plt_recorder import Recorder  # noqa: E402
from make_golden_compression import REF, lift  # noqa: E402

SEED = 77


def info_plane_inputs():
    rng = np.random.default_rng(5)
    epochs, F = 2300, 3                       # > 1000 epochs: exercises the sieve (factor 2) and the start index (500)
    kl = np.abs(rng.standard_normal((epochs, F))) * np.linspace(3.0, 0.1, epochs)[:, None]
    loss = 0.2 + 0.5 * np.exp(-np.linspace(0, 4, epochs)) + 0.01 * rng.standard_normal(epochs)
    return kl, loss


def history_inputs():
    rng = np.random.default_rng(9)
    n = 17
    h = {"beta": np.float32(np.geomspace(1e-4, 3.0, n)).tolist(), "val_beta": np.float32(np.geomspace(1e-4, 3.0, n)).tolist()}
    for f in range(3):
        h[f"KL{f}"] = (np.abs(rng.standard_normal(n)) * 2).tolist()
        h[f"val_KL{f}"] = (np.abs(rng.standard_normal(n)) * 2).tolist()
    h["loss"] = (0.3 + rng.random(n) + np.float32(h["beta"]) * sum(np.array(h[f"KL{f}"]) for f in range(3))).tolist()
    h["val_loss"] = (0.3 + rng.random(n) + np.float32(h["val_beta"]) * sum(np.array(h[f"val_KL{f}"]) for f in range(3))).tolist()
    return h


def main():
    out = {}
    kl, loss = info_plane_inputs()
    for tag, (k, l, hy) in {"long": (kl, loss, 0.758), "short": (kl[:40, :1], loss[:40], None)}.items():
        rec = Recorder()
        g = {"np": np, "plt": rec, "os": os, "default_mpl_colors": ["c%d" % i for i in range(10)]}
        fn = lift(os.path.join(REF, "visualization.py"), "save_distributed_info_plane", g)
        fn(k, l, "/tmp/unused_outdir", entropy_y=hy)
        main_plots, twin_plots = rec.plots("main"), rec.plots(("main", "twin"))
        out[f"ip_{tag}_n_main"], out[f"ip_{tag}_n_twin"] = len(main_plots), len(twin_plots)
        for i, (x, y) in enumerate(main_plots):
            out[f"ip_{tag}_main{i}_x"], out[f"ip_{tag}_main{i}_y"] = x, y
        for i, (x, y) in enumerate(twin_plots):
            out[f"ip_{tag}_twin{i}_x"], out[f"ip_{tag}_twin{i}_y"] = x, y
        out[f"ip_{tag}_saved"] = np.array(rec.saved)
    spec = importlib.util.spec_from_file_location("ref_chaos_data", os.path.join(REF, "chaos", "chaos_data.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for name, prm in (("logistic", {}), ("henon", {}), ("ikeda", {}), ("logistic_r4", {"r": 4.0})):
        np.random.seed(SEED)
        out[f"chaos_{name}"] = ref.generate_data(name.split("_")[0], number_iterations=400, number_skip_iterations=1500, **prm)
    # ---- train.py:168-178, the History post-processing statements themselves, executed on a synthetic history ----
    h = history_inputs()
    src = open(os.path.join(REF, "train.py")).read().split("\n")
    first = next(i for i, l in enumerate(src) if "beta_series = np.float32(history.history['beta'])" in l)
    last = next(i for i, l in enumerate(src) if "loss_series /= np.log(2)  ## convert the loss values to bits" in l)
    block = [l for l in src[first:last + 1]]
    indent = len(block[0]) - len(block[0].lstrip())
    code = "\n".join(l[indent:] if l.strip() else "" for l in block)
    import types
    for info_based in (True, False):
        g = {"np": np, "history": types.SimpleNamespace(history={k: list(v) for k, v in h.items()}),
             "dataset_dict": {"number_features": 3, "loss_is_info_based": info_based}}
        exec(compile(code, "train.py:%d-%d" % (first + 1, last + 1), "exec"), g)
        tag = "info" if info_based else "plain"
        out[f"hist_{tag}_loss"], out[f"hist_{tag}_kl_bits"], out[f"hist_{tag}_beta"] = g["loss_series"], g["kl_series"], g["beta_series"]
        out[f"hist_{tag}_loss_validation_raw"] = g["loss_series_validation"]   # the reference leaves it as the raw val_loss
    np.savez_compressed(os.path.join(HERE, "misc.npz"), seed=SEED, **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.endswith(("_x", "_y"))})


if __name__ == "__main__":
    main()
