"""which torch-level ops (copies, fills, small elementwise kernels) one Trainer.step still issues besides the library's launches"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deflow_amd
from torch.utils._python_dispatch import TorchDispatchMode
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4, dtype=(sys.argv[1] if len(sys.argv) > 1 else "fp32"))
b = synth_batch(int(os.environ.get("B", 16)), 80000, device=dev)
for _ in range(3):
    tr.step(b)
seen = collections.Counter()
SKIP = ("aten.empty", "aten.view", "aten.slice", "aten.select", "_unsafe_view", "aten.reshape", "aten.detach", "aten.alias", "aten.as_strided",
        "aten.unsqueeze", "aten.permute", "aten.t.", "aten.transpose", "aten.expand", "aten.squeeze", "aten.unbind", "aten._local_scalar")
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            fr = [f for f in traceback.extract_stack()[:-1] if "deflow_amd" in f.filename][-2:]
            seen[(name, " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    tr.step(b)
torch.cuda.synchronize()
print(sum(seen.values()), "torch-level ops in one step")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
