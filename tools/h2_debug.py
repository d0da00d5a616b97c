import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd._lib import img, call, ptr, stream
dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
n, h, w, cin, cout = 2, 12, 128, 64, 128
x = torch.randn(n, h, w, cin, generator=g) * 1e-3
x.view(-1)[12345] = 1e3
dy = torch.randn(n, h, w, cout, generator=g) * 1e-3
dy.view(-1)[12345] = 1e3
xd, dyd = x.to(dev), dy.to(dev)
xi, dyi = img(xd), img(dyd)
print("amax x", ops.amax_of(xi, dev).item(), float(x.abs().max()), "amax dy", ops.amax_of(dyi, dev).item(), float(dy.abs().max()))
dw = torch.empty(cout, 3, 3, cin, device=dev)
ops.conv2d_wgrad(xi, dyi, 3, 1, dw)
torch.cuda.synchronize()
bad = (~torch.isfinite(dw)).nonzero()
print("non-finite:", bad.shape[0], bad[:10].tolist())
import torch.nn.functional as F
wref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
F.conv2d(x.permute(0, 3, 1, 2).double(), wref, padding=1).backward(dy.permute(0, 3, 1, 2).double())
want = wref.grad.permute(0, 2, 3, 1)
d = (dw.cpu().double() - want)
d[~torch.isfinite(d)] = 0
print("max err among finite", float(d.abs().max() / want.abs().max()), "want max", float(want.abs().max()))
if bad.shape[0]:
    b = bad[0].tolist()
    print("want at bad", float(want[b[0], b[1], b[2], b[3]]), "got", float(dw[b[0], b[1], b[2], b[3]]))
