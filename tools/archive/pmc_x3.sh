# SQ counters of the fp32-accurate bf16x3 kernels on one layer (128 -> 128 3x3 @128^2 x 32: forward conv_halo_x3_kernel and
# weight gradient wgrad3_x3_kernel), one counter group per pass (--kernel-trace --pmc only):  gpurun -- bash tools/pmc_x3.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
n, h, cin, cout = 32, 128, 128, 128
x = torch.randn(n, h, h, cin, device=dev); w = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
y = torch.empty(n, h, h, cout, device=dev); dw = torch.empty(cout, 3, 3, cin, device=dev)
for _ in range(3):
    ops.conv2d(img(x), w, None, img(y), 3, 1)
    ops.conv2d_wgrad(img(x), img(y), 3, 1, dw)
torch.cuda.synchronize()
PY
OUT=$R/gpurun_out/${TAG}_pmc_x3.txt; : > $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_MFMA"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o p -- python /tmp/one.py > /tmp/pmc$i.log 2>&1
  DB=$(find /tmp/pmc$i -name "*.db" | head -1)
  for k in conv_halo_x3 wgrad3_x3; do python $R/tools/rocpd_pmc.py --raw $k $DB >> $OUT 2>&1 || tail -3 /tmp/pmc$i.log >> $OUT; done
done
python $R/tools/rocpd_stats.py $(find /tmp/pmc1 -name "*.db" | head -1) | grep -E "x3|kernel " >> $OUT
cat $OUT
