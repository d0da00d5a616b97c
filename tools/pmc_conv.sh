cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters.txt
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
n, h, cin, cout = 32, 128, 128, 128
x = torch.randn(n, h, h, cin, device=dev); w = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
y = torch.empty(n, h, h, cout, device=dev)
for _ in range(3):
    ops.conv2d(img(x), w, None, img(y), 3, 1)
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o p -- python /tmp/one.py > /tmp/pmc$i.log 2>&1
  python $R/tools/rocpd_pmc.py --raw conv_halo $(find /tmp/pmc$i -name "*.db" | head -1) >> $R/gpurun_out/pmc_conv.txt 2>&1 || tail -3 /tmp/pmc$i.log >> $R/gpurun_out/pmc_conv.txt
done
cat $R/gpurun_out/pmc_conv.txt
