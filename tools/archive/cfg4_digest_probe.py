"""which side is off at the configs[4] training shape (B=4, 1024x1024, 160k, 8 iters): the committed float64 digests, or the HIP
engine?  Runs the fp32 ORACLE here (no spill hooks), compares (a) its gradients' projections with the digests, (b) the HIP
gradients with the fresh oracle directly.  (tools only; the oracle is the checker)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import deflow_amd
from deflow_amd.synth import synth_batch
from deflow_amd.optim import Trainer
from oracle import ref_torch as O
from oracle.gen_digest_bs16 import project
B = int(os.environ.get("PB", "4"))
dg = dict(np.load(os.path.join(ROOT, "tests", "golden", "bs4_1024_it8_digest.npz")))
cfg = dict(voxel_size=[0.1, 0.1, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3], grid_feature_size=[1024, 1024], num_iters=8)
torch.manual_seed(46)
ref = O.DeFlow(**cfg).train()
mine = deflow_amd.DeFlow(**cfg)
mine.load_state_dict(ref.state_dict())
dev = torch.device("cuda")
mine = mine.to(dev).train()
batch = synth_batch(B, 160000, seed=20240116, grid_hw=(512, 512))
t0 = time.time()
res = ref(batch); loss = O.training_loss(res, batch); loss.backward()
print(f"oracle fp32 B={B}: loss {float(loss):.7f} (digest loss32 {float(dg['loss32']):.7f}) in {time.time() - t0:.0f} s", flush=True)
bd = {k: v.to(dev) for k, v in batch.items()}
tr = Trainer(mine, lr=0.0)
tr.flat.zero_grad(); tr.sink.begin()
lm = tr._forward_backward(bd)
torch.cuda.synchronize()
print(f"HIP loss {float(lm):.7f}")
pr = dict(ref.named_parameters())
for k, p in mine.named_parameters():
    if k.endswith(".conv.bias") and "encoder" in k:
        continue
    go, gh = pr[k].grad.double(), p.grad.detach().cpu().double()
    l2 = float(go.norm())
    e_hip = float((gh - go).norm()) / max(l2, 1e-300)
    line = f"{k:50s} HIP vs fresh oracle32 rms-rel {e_hip:.2e}"
    if B == 4:
        l2d = float(dg[f'grad.{k}.l2'])
        dpo = float(np.abs(project('grad.' + k, pr[k].grad).numpy() - dg[f'grad.{k}.proj']).max()) / l2d
        dph = float(np.abs(project('grad.' + k, p.grad).numpy() - dg[f'grad.{k}.proj']).max()) / l2d
        line += f" | vs digest: fresh oracle {dpo:.2e}, HIP {dph:.2e}"
    print(line)
