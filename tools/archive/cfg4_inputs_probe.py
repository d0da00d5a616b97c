"""hashes of the seeded inputs / initial weights of the configs[4] digest case + a few oracle gradients, to compare two machines"""
import os, sys, hashlib, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deflow_amd.synth import synth_batch
from oracle import ref_torch as O
out = sys.argv[1]
cfg = dict(voxel_size=[0.1, 0.1, 6], point_cloud_range=[-51.2, -51.2, -3, 51.2, 51.2, 3], grid_feature_size=[1024, 1024], num_iters=8)
torch.manual_seed(46)
ref = O.DeFlow(**cfg).train()
batch = synth_batch(4, 160000, seed=20240116, grid_hw=(512, 512), exact=True)
h = lambda t: hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()[:16]
d = {"threads": torch.get_num_threads(), "cpu": os.cpu_count()}
for k, v in batch.items():
    d["batch." + k] = h(torch.nan_to_num(v, nan=123.0))
for k, v in ref.state_dict().items():
    d["sd." + k] = h(v)
res = ref(batch); loss = O.training_loss(res, batch); loss.backward()
d["loss"] = float(loss.detach())
g = {k: p.grad for k, p in ref.named_parameters()}
for k in ("backbone.encoder_step_2.1.conv.weight", "backbone.encoder_step_1.0.conv.weight", "backbone.decoder_step1.u4_u5.1.weight"):
    d["grad." + k] = g[k].numpy()
    d["gradhash." + k] = h(g[k])
d["flow0"] = res["flow"][0].detach().numpy()[:1000]
np.savez(out, **d)
print({k: v for k, v in d.items() if isinstance(v, (str, int, float)) and not k.startswith("sd.")})
