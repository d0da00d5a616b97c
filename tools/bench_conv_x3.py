"""3x3 stride-1 conv layers of the fp32 training step: fp32-MFMA kernel (df_conv2d) vs the fp32-accurate bf16x3 kernel
(df_conv2d_x3), forward and data gradient; accuracy of both against float64 on one small slice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd._lib import call, img, ptr, stream
dev = torch.device("cuda")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, n, h in [(128, 128, 32, 128), (256, 128, 16, 256), (128, 128, 16, 256), (512, 256, 16, 128), (256, 256, 16, 128),
                        (64, 64, 32, 256), (128, 64, 16, 512), (64, 64, 16, 512)]:
    x = torch.randn(n, h, h, cin, device=dev); w = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y = torch.empty(n, h, h, cout, device=dev)
    gf = 2.0 * n * h * h * 9 * cin * cout / 1e9
    t32 = timeit(lambda: call("df_conv2d", img(x), ptr(w), ptr(b), img(y), 3, 1, 1, 0, 0, None, None, None, 0, stream()))
    ok = call("df_conv2d_x3_ok", img(x), img(y), 3, 1, 0, 0)
    w3 = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=dev)
    call("df_split_bf16x3", ptr(w), ptr(w3), w.numel(), stream())
    tx3 = timeit(lambda: call("df_conv2d_x3", img(x), ptr(w3), ptr(b), img(y), 3, 1, 1, 0, 0, None, None, None, 0, stream())) if ok else float("nan")
    print(f"{cin:3d}->{cout:3d} @{h}^2 x{n:2d} {gf:7.1f} GF | fp32 MFMA {t32:8.1f} us {gf / t32 * 1e3:6.0f} TF | bf16x3 {tx3:8.1f} us {gf / tx3 * 1e3:6.0f} TF-equivalent"
          f" | {t32 / tx3:.2f}x", flush=True)

print("weight gradients:")
for cin, cout, n, h in [(128, 128, 32, 128), (256, 128, 16, 256), (128, 128, 16, 256), (512, 256, 16, 128), (256, 256, 16, 128),
                        (64, 64, 32, 256), (128, 64, 16, 512), (64, 64, 16, 512), (256, 256, 32, 64)]:
    x = torch.randn(n, h, h, cin, device=dev); dy = torch.randn(n, h, h, cout, device=dev)
    gf = 2.0 * n * h * h * 9 * cin * cout / 1e9
    splits = call("df_conv2d_wgrad_splits", img(x), img(dy), 3, 1)
    ws = torch.empty(splits * cout * 9 * cin, device=dev)
    t32 = timeit(lambda: call("df_conv2d_wgrad_mp", img(x), img(dy), 3, 1, 1, ptr(ws), splits, None, 0, None, 0, stream()))
    tx3 = timeit(lambda: call("df_conv2d_wgrad_x3", img(x), img(dy), 3, 1, 1, ptr(ws), splits, None, stream()))
    print(f"{cin:3d}->{cout:3d} @{h}^2 x{n:2d} {gf:7.1f} GF | fp32 ring {t32:8.1f} us {gf / t32 * 1e3:6.0f} TF | bf16x3 {tx3:8.1f} us {gf / tx3 * 1e3:6.0f} TF-equivalent"
          f" | {t32 / tx3:.2f}x  (splits {splits})", flush=True)
