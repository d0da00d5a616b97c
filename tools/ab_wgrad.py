"""A/B micro-benchmark of the 3x3 weight-gradient kernel at bench-shaped layers"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
out = []
for (n, h, cin, cout) in [(32, 128, 128, 128), (32, 64, 256, 256), (16, 512, 64, 64), (16, 256, 256, 128)]:
    x = torch.randn(n, h, h, cin, device=dev)
    dy = torch.randn(n, h, h, cout, device=dev)
    dw = torch.empty(cout, 3, 3, cin, device=dev)
    fn = lambda: ops.conv2d_wgrad(img(x), img(dy), 3, 1, dw)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append(f"{2.0 * n * h * h * 9 * cin * cout / ms / 1e9:.1f}")
print("wgrad TF/s (incl. split reduce):", " ".join(out))
