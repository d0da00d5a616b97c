# SQ counters + HBM bytes of one bf16 3x3 layer (64->64 @512^2 x16, bias epilogue); run through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from deflow_amd._lib import DfImg, call, ptr, stream
dev = torch.device("cuda")
n, h, cin, cout = 16, 512, int(os.environ.get("CIN", "64")), int(os.environ.get("COUT", "64"))
x = torch.randn(n, h, h, cin, device=dev).bfloat16(); w = (torch.randn(cout, 3, 3, cin, device=dev) * 0.05).bfloat16()
b = torch.randn(cout, device=dev)
y = torch.empty(n, h, h, cout, device=dev, dtype=torch.bfloat16)
xi = DfImg(x.data_ptr(), n, h, h, cin, cin, n, h * h * cin, 0); yi = DfImg(y.data_ptr(), n, h, h, cout, cout, n, h * h * cout, 0)
for _ in range(3):
    call("df_conv2d_bf16", xi, ptr(w), ptr(b), yi, 3, 1, 1, 0, None, None, 0, stream())
torch.cuda.synchronize()
PY
rm -f $R/gpurun_out/pmc_bf16.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o p -- python /tmp/one.py > /tmp/pmc$i.log 2>&1
  python $R/tools/rocpd_pmc.py --raw conv_halo_bf16 $(find /tmp/pmc$i -name "*.db" | head -1) >> $R/gpurun_out/pmc_bf16.txt 2>&1 || tail -3 /tmp/pmc$i.log >> $R/gpurun_out/pmc_bf16.txt
done
cat $R/gpurun_out/pmc_bf16.txt
