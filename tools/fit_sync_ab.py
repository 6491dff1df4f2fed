"""Same-box A/B of fit's epoch boundary on the reference-default run (Boolean circuit, F = 10, B = 128, validation every epoch):
one synchronisation per epoch (validation sums in an accumulator of their own, read together with the training sums) against
round 5's two (`model.syncs_per_epoch = 2`).  usage: python tools/fit_sync_ab.py [epochs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dib_amd  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 400
d = dib_amd.data.fetch_boolean_circuit()
for rep in range(3):
    for syncs in (2, 1):
        m = dib_amd.DistributedIBNet(d['feature_dimensionalities'], [128, 128], [256, 256], 1)
        m.syncs_per_epoch = syncs
        opt = dib_amd.optimizers.get('adam')
        opt.learning_rate = 3e-4
        m.compile(optimizer=opt, loss=d['loss'], metrics=d['metrics'])
        cb = dib_amd.InfoBottleneckAnnealingCallback(1e-4, 3.0, 10, 40)
        kw = dict(batch_size=128, callbacks=[cb], verbose=False, validation_data=(d['x_valid'], d['y_valid']))
        m.fit(d['x_train'], d['y_train'], epochs=5, **kw)
        torch.cuda.synchronize()
        t = time.time()
        m.fit(d['x_train'], d['y_train'], epochs=E, **kw)
        torch.cuda.synchronize()
        el = time.time() - t
        print(f"syncs per epoch {syncs}: {el / E * 1e3:.3f} ms/epoch = {el / E / 8 * 1e6:.1f} us per train+val step pair", flush=True)
