"""stride-2 3x3 weight gradient at the bench's layers, fp32 and bf16-operand mode (DF_WGRAD_RING_S2=1: the 12-wave ring kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
for bf in (False, True):
    out = []
    with ops.mfma_bf16(bf):
        for (n, h, cin, cout) in [(32, 256, 64, 128), (32, 128, 128, 256), (32, 512, 32, 64)]:
            x = torch.randn(n, h, h, cin, device=dev)
            dy = torch.randn(n, h // 2, h // 2, cout, device=dev)
            dw = torch.empty(cout, 3, 3, cin, device=dev)
            fn = lambda: ops.conv2d_wgrad(img(x), img(dy), 3, 2, dw)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            if not bf and h <= 128:   # correctness of the form under test against autograd
                xr = x.permute(0, 3, 1, 2).requires_grad_(False)
                w = torch.zeros(cout, cin, 3, 3, device=dev, requires_grad=True)
                F.conv2d(xr, w, None, stride=2, padding=1).backward(dy.permute(0, 3, 1, 2))
                err = float((dw.permute(0, 3, 1, 2) - w.grad).abs().max() / w.grad.abs().max())
                out.append(f"err {err:.1e}")
            out.append(f"{2.0 * n * (h // 2) ** 2 * 9 * cin * cout / ms / 1e9:.0f} TF ({ms * 1e3:.0f} us)")
            del x, dy, dw
    print("bf16" if bf else "fp32", "s2 wgrad:", " | ".join(out))
