# SQ counters of one 3x3 weight-gradient layer (128->128 @128^2 x32); run through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
n, h, cin, cout = 32, 128, 128, 128
x = torch.randn(n, h, h, cin, device=dev); dy = torch.randn(n, h, h, cout, device=dev)
dw = torch.empty(cout, 3, 3, cin, device=dev)
for _ in range(3):
    ops.conv2d_wgrad(img(x), img(dy), 3, 1, dw)
torch.cuda.synchronize()
PY
rm -f $R/gpurun_out/pmc_wgrad.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o p -- python /tmp/one.py > /tmp/pmc$i.log 2>&1
  python $R/tools/rocpd_pmc.py --raw wgrad3_ring $(find /tmp/pmc$i -name "*.db" | head -1) >> $R/gpurun_out/pmc_wgrad.txt 2>&1 || tail -3 /tmp/pmc$i.log >> $R/gpurun_out/pmc_wgrad.txt
done
cat $R/gpurun_out/pmc_wgrad.txt
