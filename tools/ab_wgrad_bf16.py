"""3x3 weight-gradient kernel at bench-shaped layers, fp32 and bf16-operand mode (TFLOP/s incl. the split reduce)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
SHAPES = [(32, 128, 128, 128), (32, 64, 256, 256), (16, 512, 64, 64), (16, 256, 256, 128), (16, 128, 512, 256), (16, 512, 128, 64),
          (32, 256, 64, 64)]
for bf in (False, True):
    out = []
    tot = 0.0
    with ops.mfma_bf16(bf):
        for (n, h, cin, cout) in SHAPES:
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
            tot += ms
            out.append(f"{2.0 * n * h * h * 9 * cin * cout / ms / 1e9:.0f}")
            del x, dy, dw
    print("bf16" if bf else "fp32", "wgrad TF/s:", " ".join(out), f" sum {tot:.3f} ms")
