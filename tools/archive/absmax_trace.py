"""which descriptors of a B = 1 evaluation forward still cost a df_absmax pass (no producer left a bound)?"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from deflow_amd import ops, _lib
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).eval()
b = synth_batch(int(os.environ.get("B", 1)), 80000, device=dev)
with torch.no_grad():
    for _ in range(3):
        m.forward_padded(b)
seen = collections.Counter()
real = ops.call
def spy(name, *a):
    if name == "df_absmax":
        d = a[0]
        fr = [f for f in traceback.extract_stack()[:-1] if "deflow_amd" in f.filename][-4:]
        seen[(f"{d.n}x{d.h}x{d.w}x{d.c} ld{d.ld}", " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
    return real(name, *a)
ops.call = spy
with torch.no_grad():
    m.forward_padded(b)
torch.cuda.synchronize()
for k, v in seen.most_common():
    print(v, k)
