import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn.functional as F
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
for cin, cout, n, h, w in [(32, 64, 3, 16, 16), (32, 64, 2, 16, 16), (64, 128, 3, 16, 16), (64, 64, 5, 12, 20), (32, 64, 1, 10, 14), (128, 256, 3, 8, 8)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev); gyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = ops.ohwi(wt.detach().to(dev).contiguous(memory_format=torch.channels_last))
    dx = torch.empty_like(xd)
    ops.conv2d(img(gyd), ops.weight_transpose(wd), None, img(dx), 3, 2, mode=ops.CONV_DGRAD)
    e = float((dx.permute(0, 3, 1, 2).cpu() - x.grad).abs().max() / x.grad.abs().max())
    print(cin, cout, n, h, w, "dgrad s2 rel err", e)
