"""per-layer time of the 3x3 stride-1 convolutions of the B = 16 training step in fp32 mode: fp16x2 (default) vs bf16x3
(DF_CONV_H2=0) forms, forward / data gradient / weight gradient.  usage: python tools/bench_conv_h2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
LAYERS = [(64, 64, 32, 256, 256), (128, 128, 32, 128, 128), (256, 256, 32, 64, 64), (512, 256, 16, 128, 128), (256, 256, 16, 128, 128),
          (256, 128, 16, 256, 256), (128, 128, 16, 256, 256), (128, 64, 16, 512, 512), (64, 64, 16, 512, 512)]
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cin, cout, n, h, w in LAYERS:
    x = torch.randn(n, h, w, cin, device=dev); dy = torch.randn(n, h, w, cout, device=dev)
    wt = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
    y = torch.empty(n, h, w, cout, device=dev); dw = torch.empty(cout, 3, 3, cin, device=dev)
    fl = 2.0 * n * h * w * 9 * cin * cout
    row = f"{cin:3d}->{cout:3d} @{h}x{w}x{n}: "
    for form in ("1", "0"):
        os.environ["DF_CONV_H2"] = form
        xi, dyi = img(x), img(dy)          # (descriptor objects carry the measured amax: taken once, outside the timing, as in a step)
        ops.amax_of(xi, dev); ops.amax_of(dyi, dev)
        tf = t(lambda: ops.conv2d(xi, wt, None, img(y), 3, 1))
        tw = t(lambda: ops.conv2d_wgrad(xi, dyi, 3, 1, dw))
        row += f"{'h2' if form == '1' else 'x3'} fwd {tf:.3f} ms {fl / tf / 1e9:6.1f} TF  wgrad {tw:.3f} ms {fl / tw / 1e9:6.1f} TF | "
    ta = t(lambda: ops.amax_of(img(x), dev))
    print(row + f"absmax(x) {ta * 1e3:.0f} us")
