"""bf16 conv prototype: correctness vs torch (bf16-rounded inputs, fp32 accumulate) and throughput"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deflow_amd._lib import DfImg, call, ptr, stream
dev = torch.device("cuda")
def run(n, h, cin, cout, ks=3, stride=1, check=False):
    x = torch.randn(n, h, h, cin, device=dev).bfloat16()
    w = (torch.randn(cout, ks, ks, cin, device=dev) * 0.05).bfloat16()
    b = torch.randn(cout, device=dev)
    ho = (h + 2 * (ks // 2) - ks) // stride + 1
    y = torch.empty(n, ho, ho, cout, device=dev, dtype=torch.bfloat16)
    xi = DfImg(x.data_ptr(), n, h, h, cin, cin, n, h * h * cin, 0)
    yi = DfImg(y.data_ptr(), n, ho, ho, cout, cout, n, ho * ho * cout, 0)
    fn = lambda: call("df_conv2d_bf16", xi, ptr(w), ptr(b), yi, ks, stride, ks // 2, 0, None, None, 0, stream())
    fn(); torch.cuda.synchronize()
    if check:
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride, ks // 2).permute(0, 2, 3, 1)
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        print(f"  check {cin}->{cout} k{ks} s{stride} @{h}: max rel err {err:.2e} (bf16 output rounding ~4e-3)")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"bf16 conv {cin}->{cout} k{ks} s{stride} @{h}^2 x{n}: {ms:.3f} ms  {2.0 * n * ho * ho * ks * ks * cin * cout / ms / 1e9:.0f} TF/s")
run(2, 32, 64, 64, check=True); run(2, 32, 128, 128, check=True); run(2, 32, 64, 128, 3, 2, check=True); run(2, 32, 128, 64, 1, 1, check=True)
run(2, 128, 64, 64, check=True); run(2, 128, 128, 128, check=True); run(1, 256, 192, 64, check=True)   # haloed kernels (W % 128 == 0)
for (n, h, cin, cout) in [(32, 128, 128, 128), (32, 256, 64, 64), (16, 512, 64, 64), (16, 512, 128, 64), (16, 256, 256, 128), (32, 128, 256, 256)]:
    run(n, h, cin, cout)
