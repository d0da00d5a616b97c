"""Feasibility / gain of replaying the whole training step (forward, loss, hand-sequenced backward, Adam) as ONE captured
HIP graph: usage python tools/graph_step.py [bf16|fp32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4, dtype=dtype)
b = synth_batch(16, 80000, device=dev)


def timeit(fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        tr.step(b)
torch.cuda.current_stream().wait_stream(s)
eager = timeit(lambda: tr.step(b))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = tr.step(b)
g.replay(); torch.cuda.synchronize()
rep = timeit(g.replay)
t0 = time.perf_counter(); g.replay(); host = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
print(f"{dtype}: eager {eager:.2f} ms/step, graph replay {rep:.2f} ms/step (host time per replay {host:.2f} ms), loss {float(loss):.4f}")
