python tools/bench_infer.py
cd /tmp && export TMPDIR=/tmp
cat > /tmp/inf1.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, deflow_amd
from deflow_amd.synth import synth_batch
dev = torch.device("cuda"); torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).eval()
batch = synth_batch(1, 80000, device=dev)
with torch.no_grad():
    for _ in range(13):
        m.forward_padded(batch)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats -d /tmp/ki -o ki -- python /tmp/inf1.py > /tmp/ki.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/ki -name "*.db" | head -1) | head -30
