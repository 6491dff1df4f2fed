"""Pin of the custom-loop ACCOUNTING (reference train.py:222-282) obtained by EXECUTING the reference's own statements
(build container only; numbers stored, no source copied):

    python tests/golden/make_golden_infonce_loop.py   ->   tests/golden/infonce_loop.npz

The statements train.py:224-225 (steps_per_epoch), 230-231 (number_full_validation_batches) and 236-285 (epoch_steps, the
step loop with its boundary test, the numpy beta formula, model.beta.assign, the validation loop, the epoch means, the
final conversions) are cut out of the source by line content, dedented and exec'd.  What they call is stubbed:
  * tf_dataset.take(n)              -> n (step index, None) pairs                      [tf.data: not executable here]
  * tf_dataset_validation           -> number_full_validation_batches + 1 items per iteration (what train.py:234's
                                       `.take(number_full_validation_batches+1)` yields; restated, tf.data again)
  * eval_batch_infonce(i, o, training) -> a deterministic function of (call number, training flag, CURRENT model.beta),
                                       so the series also pin WHEN beta changes relative to the steps (the first step runs
                                       at the constructor's beta = 1, models.py:86)
  * model.beta                      -> float32 cell with assign()/value()
  * save_compression_matrices_frequency = 0 (the figure branch is pinned elsewhere).
oracle/infonce_loop_oracle.run_loop must give the same series from the same stubs (tests/test_oracle_golden.py)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

CASES = {  # name: (dataset_length, validation_set_length, batch_size, n_pre, n_anneal, beta_start, beta_end, kl width)
    "fractional": (1000, 300, 128, 3, 5, 1e-4, 3.0, 3),
    "bankers_half": (320, 64, 128, 2, 6, 1e-3, 1.0, 1),       # steps_per_epoch = 2.5: round(2.5) = 2, round(7.5) = 8
    "repeated_boundaries": (100, 40, 128, 1, 8, 1e-2, 2.0, 4),  # steps_per_epoch < 1: several epochs share a step
    "exact": (1024, 256, 128, 0, 4, 1e-4, 3.0, 2),
}


def stub_values(call: int, training: bool, beta: float, width: int):
    """(loss_infonce, kl vector) of call number `call` - shared with the test (imported from here)."""
    base = np.sin(0.37 * call + (0.0 if training else 1.3))
    loss = 2.0 + base + 0.01 * float(beta)
    kl = np.abs(np.cos(0.11 * call * (np.arange(width) + 1.0))) * (1.0 + 0.5 * float(beta)) + (0.0 if training else 0.25)
    return loss, kl


def lifted_code():
    src = open(os.path.join(REF, "train.py")).read().split("\n")

    def find(s, start=0):
        return next(i for i in range(start, len(src)) if s in src[i])

    pieces = []
    a = find("dataset_length = dataset_dict['x_train'].shape[0]")
    pieces += src[a: a + 2]                                                     # 224-225
    b = find("validation_set_length = dataset_dict['x_valid'].shape[0]")
    pieces += src[b: b + 2]                                                     # 230-231
    pieces.append("    tf_dataset_validation = make_validation_dataset(number_full_validation_batches + 1)")  # 233-234 restated
    c = find("epoch_steps = np.round(steps_per_epoch*np.arange(number_epochs)).astype(np.int32)")
    d = find("kl_series_validation /= np.log(2)", c)
    pieces += src[c: d + 1]                                                     # 236-285
    indent = 4
    return "\n".join(l[indent:] if l.strip() else "" for l in pieces), (a + 1, d + 1)


class _Beta:
    def __init__(self):
        self.v = np.float32(1.0)       # models.py:86

    def assign(self, x):
        self.v = np.float32(x)

    def value(self):
        return self.v


def run_reference(case):
    n, nv, bs, n_pre, n_ann, b0, b1, width = case
    code, span = lifted_code()
    beta = _Beta()
    calls = {"n": 0}
    log = []

    def eval_batch_infonce(inps, outps, training=True):
        l, k = stub_values(calls["n"], training, beta.value(), width)
        log.append((calls["n"], int(training), float(beta.value())))
        calls["n"] += 1
        return l, k

    class _Train:
        def take(self, k):
            return [(i, None) for i in range(int(k))]

    class _Valid:
        def __init__(self, k):
            self.k = k

        def __iter__(self):
            return iter([(None, None)] * self.k)

    import types
    g = dict(np=np, batch_size=bs, number_epochs=n_pre + n_ann, number_pretraining_epochs=n_pre, number_annealing_epochs=n_ann,
             beta_start=b0, beta_end=b1, save_compression_matrices_frequency=0,
             dataset_dict={"x_train": np.zeros((n, 1)), "x_valid": np.zeros((nv, 1)), "loss_is_info_based": False},
             tf_dataset=_Train(), make_validation_dataset=_Valid, eval_batch_infonce=eval_batch_infonce,
             model=types.SimpleNamespace(beta=beta))
    exec(compile(code, "train.py:%d-%d" % span, "exec"), g)
    return dict(beta=g["beta_series"], kl_bits=g["kl_series"], loss=g["loss_series"],              # KL already /= ln 2 (l.284-285)
                kl_bits_validation=g["kl_series_validation"], loss_validation=g["loss_series_validation"],
                epoch_steps=g["epoch_steps"], calls=np.array(log, dtype=np.float64))


def main():
    out = {}
    for name, case in CASES.items():
        r = run_reference(case)
        for k, v in r.items():
            out[f"{name}_{k}"] = np.asarray(v)
        print(name, {k: np.asarray(v).shape for k, v in r.items()})
    np.savez_compressed(os.path.join(HERE, "infonce_loop.npz"), **out)


if __name__ == "__main__":
    main()
