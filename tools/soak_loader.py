"""Soak test of loader-fed training over two epochs: throughput, host RSS and GPU memory over time (run through gpurun)."""
import os, sys, time, subprocess
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
root = "/tmp/df_bigscenes"
if not os.path.exists(os.path.join(root, "index_total.pkl")):
    import runpy
    # reuse the generator of tools/bench_loader.py by executing it once with a tiny step budget is overkill: call its GEN
    src = open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools", "bench_loader.py")).read()
    gen = src.split("GEN = r'''")[1].split("'''")[0]
    os.makedirs(root, exist_ok=True)
    subprocess.run(["/opt/conda/bin/python3.9", "-c", gen, root], check=True)
import torch, deflow_amd, psutil
from deflow_amd.data import HDF5Dataset, SceneLoader, ShardedSampler
from deflow_amd.optim import Trainer
dev = torch.device("cuda")
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4)
ds = HDF5Dataset(root); ds.data_index = ds.data_index * 6
proc = psutil.Process()
t0 = time.perf_counter(); n = 0
s = ShardedSampler(len(ds), shuffle=True, seed=1)
loader = SceneLoader(ds, 16, s, device=dev, num_workers=4)
for ep in range(2):
    s.set_epoch(ep)
    for k, b in enumerate(loader):
        loss = tr.step(b); n += 16
        if k % 20 == 0:
            torch.cuda.synchronize()
            print(f"ep {ep} step {k} loss {float(loss.detach())/16:.4f} host RSS {proc.memory_info().rss/2**30:.2f} GiB, gpu alloc {torch.cuda.memory_allocated()/2**30:.2f} reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB, {n/(time.perf_counter()-t0):.1f} pairs/s", flush=True)
torch.cuda.synchronize()
print("done", n, "pairs", f"{n/(time.perf_counter()-t0):.1f} pairs/s overall")
