import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).eval()
b = synth_batch(1, 80000, device=dev)
with torch.no_grad():
    for _ in range(12):
        m.forward_padded(b)
torch.cuda.synchronize()
