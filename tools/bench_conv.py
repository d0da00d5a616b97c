"""micro-benchmark of conv_kernel at a bench-shaped layer: 128->128 3x3 on [32,128,128] (154.6 GFLOP)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
for (n, h, cin, cout) in [(32, 128, 128, 128), (32, 64, 256, 256), (16, 512, 64, 64), (16, 256, 256, 128)]:
    x = torch.randn(n, h, h, cin, device=dev)
    w = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
    y = torch.empty(n, h, h, cout, device=dev)
    fn = lambda: ops.conv2d(img(x), w, None, img(y), 3, 1)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"conv {cin}->{cout} @{h}^2 x{n}: {ms:.3f} ms  {2.0 * n * h * h * 9 * cin * cout / ms / 1e9:.1f} TF/s")
