#!/bin/bash
# r04q: the paper circuit at train.py's EXACT default schedule (1000 + 10000 epochs, B = 128, lr 3e-4, beta 1e-4 -> 3): wall time and
# drop order against the reference notebook's printed TensorFlow outcome
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r04q; mkdir -p $O
timeout 600 python - > $O/paper_circuit_defaults.txt 2> $O/err.txt <<'PY'
import sys, time, importlib.util, numpy as np
spec = importlib.util.spec_from_file_location("pcr", "tools/paper_circuit_run.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
for seed in (0, 1):
    t0 = time.time()
    drop, kl, loss, acc, beta = m.run(epochs_pre=1000, epochs_anneal=10000, seed=seed, learning_rate=3e-4, beta_start=1e-4, beta_end=3.0)
    print(f"train.py defaults, seed {seed}: {time.time() - t0:.1f} s for 11000 epochs; drop epochs", drop.tolist(), "order",
          np.argsort(drop, kind="stable").tolist(), "violations", m.group_order_violations(drop),
          f"acc@pre {acc[999]:.3f} loss@pre {loss[999]:.3f} max acc {acc.max():.3f} final KL {kl[-1].sum():.3f} final loss {loss[-1]:.3f} bits", flush=True)
PY
cat $O/paper_circuit_defaults.txt; tail -n 3 $O/err.txt
