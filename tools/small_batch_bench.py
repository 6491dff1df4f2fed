"""Launch-bound regime check: the reference default Boolean-circuit run (train.py:30-34: B=128, 8 steps/epoch)."""
import time, numpy as np, torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dib_amd
from dib_amd import _lib
for kv in [a for a in sys.argv[1:] if '=' in a]:   # dib_set_tuning keys, e.g. int_cluster=0
    _lib.set_tuning(kv.split('=')[0], int(kv.split('=')[1]))
d=dib_amd.data.fetch_boolean_circuit()
m=dib_amd.DistributedIBNet(d['feature_dimensionalities'],[128,128],[256,256],1)
opt=dib_amd.optimizers.get('adam'); opt.learning_rate=3e-4
m.compile(optimizer=opt, loss=d['loss'], metrics=d['metrics'])
cb=dib_amd.InfoBottleneckAnnealingCallback(1e-4,3.0,10,40)
m.fit(d['x_train'],d['y_train'],epochs=3,batch_size=128,callbacks=[cb],verbose=False,validation_data=(d['x_valid'],d['y_valid']))
torch.cuda.synchronize(); t=time.time()
E=int(os.environ.get("DIB_SMALL_EPOCHS", "300"))
m.fit(d['x_train'],d['y_train'],epochs=E,batch_size=128,callbacks=[cb],verbose=False,validation_data=(d['x_valid'],d['y_valid']))
torch.cuda.synchronize(); el=time.time()-t
print(" ".join(sys.argv[1:]), f"boolean circuit (F=10, B=128): {el/E*1e3:.2f} ms/epoch (8 train + 8 val steps) -> {el/E/8*1e6:.1f} us per train+val step pair; 11000 epochs = {el/E*11000:.0f} s")
