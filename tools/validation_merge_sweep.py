"""fit on the reference's default run (Boolean circuit, F = 10, B = 128, 8 validation batches per epoch) for several values of
model.validation_merge_rows (how many rows of full validation batches one launch set evaluates; 0 = batch by batch):
microseconds per (training + validation) step pair.  profiles/r05zz_validation_merge_rows_sweep.txt."""
import sys
import time

import torch
sys.path.insert(0, '.')
import bench, dib_amd
d = dib_amd.data.fetch_boolean_circuit()
m = dib_amd.DistributedIBNet(d["feature_dimensionalities"], bench.ENC, bench.INTEG, 1, feature_embedding_dimension=bench.E, device='cuda:0')
opt = dib_amd.optimizers.get("adam"); opt.learning_rate = 3e-4
m.compile(optimizer=opt, loss=d["loss"], metrics=d["metrics"])
cb = dib_amd.InfoBottleneckAnnealingCallback(1e-4, 3.0, 10, 40)
kw = dict(batch_size=128, callbacks=[cb], verbose=False, validation_data=(d["x_valid"], d["y_valid"]))
for rep in range(2):
    for rows in (1024, 768, 512, 256, 0):
        m.validation_merge_rows = rows
        m.fit(d["x_train"], d["y_train"], epochs=3, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.fit(d["x_train"], d["y_train"], epochs=200, **kw)
        torch.cuda.synchronize()
        print("validation_merge_rows", rows, "us per pair", round((time.perf_counter() - t0) / 200 / 8 * 1e6, 1), flush=True)
