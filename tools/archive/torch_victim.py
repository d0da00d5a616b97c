"""Victim made of PyTorch's OWN kernels (block reductions through LDS, a sort, a small GEMM): are THEY bit-reproducible while
another process runs the bf16 training loop on the same GPU?  (If not, the cross-process interference is the platform's,
not this repo's kernels'.)   python tools/torch_victim.py [reps]"""
import sys
import torch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(2048, 288, generator=g).to(dev)
k = torch.randint(0, 1 << 20, (20000,), generator=g).to(dev)
a = torch.randn(256, 256, generator=g).to(dev)
first, bad = None, {}
for r in range(reps):
    outs = {"sum0": x.sum(0), "sum1": x.sum(1), "var": x.var(0), "softmax": torch.softmax(x, 1), "sort": torch.sort(k).values.float(),
            "cumsum": torch.cumsum(x, 0), "mm": a @ a, "layer_norm": torch.nn.functional.layer_norm(x, (288,))}
    torch.cuda.synchronize()
    if first is None:
        first = {n: v.clone() for n, v in outs.items()}
        continue
    for n, v in outs.items():
        if not torch.equal(v, first[n]):
            bad[n] = bad.get(n, 0) + 1
print(f"torch-kernel victim: reps={reps} differing: {bad if bad else 'none'}")
