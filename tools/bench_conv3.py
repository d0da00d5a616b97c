"""3x3 stride-1 forward conv, three ways: fp32 MFMA, bf16-operand mode on fp32 tensors, bf16 storage kernel (inference path).
Shapes from the training bench (B=16) and from the configs[4] inference forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd import ops
from deflow_amd._lib import DfImg, call, img, ptr, stream
dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, n, h in [(128, 128, 32, 128), (128, 128, 2, 256), (256, 256, 2, 128), (64, 64, 2, 512), (64, 64, 1, 1024), (256, 128, 1, 256),
                        (128, 128, 2, 64), (256, 256, 2, 32)]:
    x = torch.randn(n, h, h, cin, device=dev); w = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y = torch.empty(n, h, h, cout, device=dev)
    gf = 2.0 * n * h * h * 9 * cin * cout / 1e9
    t32 = timeit(lambda: ops.conv2d(img(x), w, b, img(y), 3, 1))
    with ops.mfma_bf16(True):
        tmp = timeit(lambda: ops.conv2d(img(x), w, b, img(y), 3, 1))
    x16, w16 = x.bfloat16(), w.bfloat16()
    y16 = torch.empty(n, h, h, cout, device=dev, dtype=torch.bfloat16)
    xi = DfImg(x16.data_ptr(), n, h, h, cin, cin, n, h * h * cin, 0); yi = DfImg(y16.data_ptr(), n, h, h, cout, cout, n, h * h * cout, 0)
    t16 = timeit(lambda: call("df_conv2d_bf16", xi, ptr(w16), ptr(b), yi, 3, 1, 1, 0, None, None, 0, stream()))
    print(f"{cin:3d}->{cout:3d} @{h}^2 x{n:2d} {gf:7.1f} GF | fp32 {t32:8.1f} us {gf / t32 * 1e3:6.0f} TF | bf16 operands {tmp:7.1f} us {gf / tmp * 1e3:6.0f} TF | "
          f"bf16 storage {t16:7.1f} us {gf / t16 * 1e3:6.0f} TF")
