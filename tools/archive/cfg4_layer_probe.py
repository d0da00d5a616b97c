"""per-module forward outputs / backward grad_outputs of the ORACLE at the configs[4] digest case, as (sum, l2, a fixed projection):
run on two hosts and diff -- where do two CPU runs of the same fp32 graph part ways?  (tools only)
    python tools/cfg4_layer_probe.py out.npz [threads]"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deflow_amd.synth import synth_batch
from oracle import ref_torch as O
out = sys.argv[1]
if len(sys.argv) > 2:
    torch.set_num_threads(int(sys.argv[2]))
cfg = dict(voxel_size=[0.1, 0.1, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3], grid_feature_size=[1024, 1024], num_iters=8)
torch.manual_seed(46)
ref = O.DeFlow(**cfg).train()
batch = synth_batch(4, 160000, seed=20240116, grid_hw=(512, 512), exact=True)
rec = {}
def digest(t):
    t = t.detach().double().reshape(-1)
    n = t.numel()
    idx = torch.arange(min(n, 1 << 22), dtype=torch.float64)
    w = torch.cos(idx * 0.37)
    return np.array([float(t.sum()), float(t.norm()), float((t[:w.numel()] * w).sum()), float(n)])
def fwd_hook(name):
    def h(m, i, o):
        if isinstance(o, torch.Tensor):
            rec.setdefault("fwd." + name, []).append(digest(o))
            if o.requires_grad:
                o.register_hook(lambda g, name=name: rec.setdefault("bwd." + name, []).append(digest(g)))
    return h
for name, m in ref.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_hook(fwd_hook(name))
res = ref(batch); loss = O.training_loss(res, batch); loss.backward()
d = {k: np.stack(v) for k, v in rec.items()}
for k, p in ref.named_parameters():
    d["grad." + k] = digest(p.grad)
d["threads"] = torch.get_num_threads()
np.savez(out, **d)
print("loss", float(loss.detach()), "threads", torch.get_num_threads(), len(d))
