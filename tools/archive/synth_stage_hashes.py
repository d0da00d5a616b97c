"""per-stage sha256 of deflow_amd.synth.synth_pair(exact=True)'s intermediates: which operation differs between two hosts?"""
import hashlib, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd.synth import _rigid, _sqrt_f32
h = lambda t: hashlib.sha256(torch.nan_to_num(t.float(), nan=7.0).contiguous().numpy().tobytes()).hexdigest()[:12]
seed, n = 20240116, 160000
g = torch.Generator().manual_seed(seed)
xy = torch.randn(n, 2, generator=g) * 20.0; print("xy", h(xy))
z = (torch.rand(n, 1, generator=g) * 6.6) - 3.3; print("z", h(z))
pc0 = torch.cat([xy, z], 1)
yaw = (torch.rand(1, generator=g).item() * 4 - 2) * math.pi / 180
T = torch.eye(4)
T[0, 0] = math.cos(yaw); T[0, 1] = -math.sin(yaw); T[1, 0] = math.sin(yaw); T[1, 1] = math.cos(yaw)
T[0, 3] = torch.rand(1, generator=g).item() * 1.5
print("T", h(T), repr(yaw))
dyn = torch.rand(n, generator=g) < 0.1; print("dyn", h(dyn))
d = torch.randn(n, 3, generator=g); print("d raw", h(d))
sq = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]; print("sumsq", h(sq))
nrm = _sqrt_f32(sq).unsqueeze(1); print("nrm", h(nrm)); print("torch.sqrt", h(torch.sqrt(sq)))
q = d / nrm; print("d/nrm", h(q))
u = torch.rand(n, 1, generator=g) * 2.0; print("u", h(u))
d2 = q * u; print("d*u", h(d2))
moved = _rigid(pc0, T); print("moved", h(moved))
noise = torch.randn(n, 3, generator=g) * 0.02; print("noise", h(noise))
print("cos/sin", repr(math.cos(yaw)), repr(math.sin(yaw)))
