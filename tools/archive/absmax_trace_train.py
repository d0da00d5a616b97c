"""which descriptors of a training step still cost a df_absmax pass (no producer left a bound)?"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd import ops
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4)
b = synth_batch(16, 80000, device=dev)
for _ in range(2):
    tr.step(b)
seen = collections.Counter()
import deflow_amd.optim as O
real = ops.call
def spy(name, *a):
    if name == "df_absmax":
        d = a[0]
        fr = [f for f in traceback.extract_stack()[:-1] if "deflow_amd" in f.filename][-5:]
        seen[(f"{d.n}x{d.h}x{d.w}x{d.c} ld{d.ld}", " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
    return real(name, *a)
ops.call = spy
O.call = spy
tr.step(b)
torch.cuda.synchronize()
for k, v in seen.most_common():
    print(v, k)
