"""which torch-level ops (copies, fills, small elementwise kernels) a B = 1 evaluation forward still issues besides the library's launches"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from torch.utils._python_dispatch import TorchDispatchMode
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).eval()
b = synth_batch(1, 80000, device=dev)
with torch.no_grad():
    for _ in range(3):
        m.forward_padded(b)
seen = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        fr = [f for f in traceback.extract_stack()[:-1] if "deflow_amd" in f.filename][-3:]
        seen[(str(func), " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))
with torch.no_grad(), Spy():
    m.forward_padded(b)
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
