# Round profile artefacts on the GPU box (run through gpurun; results land in gpurun_out/<TAG>_*; copy what is to be judged
# into profiles/).  usage: bash tools/prof_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of `python bench.py` (the DEFAULT command: training region + the forward-only /
#      bf16 extras) -> <TAG>_full_kernel_stats.txt, <TAG>_bench.json (the JSON line printed under the profiler)
#   2. the same with --no-extras --no-cpu-baseline: training launches only -> <TAG>_train_kernel_stats.txt (this is the table
#      whose per-kernel averages must agree with the HIP-event figures in the bench line's `roofline`)
#   3. the training step with dtype=bf16 only -> <TAG>_bf16_train_kernel_stats.txt
#   4. two separate PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as the pool requires) -> HBM bytes per launch
#   5. SQ counter passes of the dominant conv kernel, fp32 and bf16-operand mode -> <TAG>_pmc_conv_{f32,bf16}.txt
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
stats() { python $R/tools/rocpd_stats.py $(find $1 -name "*.db" | head -1); }
DF_BENCH_NO_SUBPROC=1 rocprofv3 --kernel-trace --stats -d /tmp/kt_full -o kt -- python $R/bench.py --no-loader > /tmp/kt_full.log 2>&1   # (the fresh-process legs -- strict_fp32, gru_fp32, loader_fed, the MFMA probe -- are not traced)
stats /tmp/kt_full > $O/${TAG}_full_kernel_stats.txt 2>&1
grep "^{\"metric\"" /tmp/kt_full.log | tail -1 > $O/${TAG}_bench.json
rocprofv3 --kernel-trace --stats -d /tmp/kt_train -o kt -- python $R/bench.py --no-extras --no-cpu-baseline > /tmp/kt_train.log 2>&1
stats /tmp/kt_train > $O/${TAG}_train_kernel_stats.txt 2>&1
grep "^{\"metric\"" /tmp/kt_train.log | tail -1 > $O/${TAG}_train_bench.json
cat > /tmp/bf16_train.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, deflow_amd
from deflow_amd.optim import Trainer
from deflow_amd.synth import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
m = deflow_amd.DeFlow().to(dev).train()
tr = Trainer(m, lr=2e-4, dtype="bf16")
b = synth_batch(16, 80000, device=dev)
for _ in range(2):
    tr.step(b)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(5):
    tr.step(b)
torch.cuda.synchronize()
print(f"bf16 train step: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
PY
rocprofv3 --kernel-trace --stats -d /tmp/kt_bf -o kt -- python /tmp/bf16_train.py > /tmp/kt_bf.log 2>&1
(grep "bf16 train step" /tmp/kt_bf.log; stats /tmp/kt_bf) > $O/${TAG}_bf16_train_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
python $R/tools/rocpd_pmc.py $F $W > $O/${TAG}_pmc_hbm_bytes.txt 2>&1
python $R/tools/rocpd_pmc.py --json $O/pmc_traffic.json $F $W
# SQ counters of the dominant conv layer (128->128 3x3 @128^2 x32: forward + weight gradient), fp32 mode (the fp16x2 kernels
# conv_halo_x3_kernel<..,2> / wgrad3_x3_kernel<2>) and bf16 training mode (bf16 storage: conv_halo_w16_kernel<..,true> / wgrad3_tr_kernel)
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from deflow_amd import ops
from deflow_amd._lib import img
dev = torch.device("cuda")
bf = os.environ.get("BF") == "1"
dt = torch.bfloat16 if bf else torch.float32
n, h, cin, cout = 32, 128, 128, 128
x = torch.randn(n, h, h, cin, device=dev).to(dt); w = torch.randn(cout, 3, 3, cin, device=dev) * 0.05
y = torch.empty(n, h, h, cout, device=dev, dtype=dt)
dy = torch.randn(n, h, h, cout, device=dev).to(dt); dw = torch.empty(cout, 3, 3, cin, device=dev)
if not bf and os.environ.get("DF_H2P", "1") != "0":      # round 4: the fp32 step's layers read PRE-SPLIT planes (conv_halo_x3_kernel<..,XP>, wgrad3_h2p_kernel)
    xb = torch.tensor([float(x.abs().max()) * 1.5], device=dev); yb = torch.tensor([float(dy.abs().max()) * 1.5], device=dev)
    x, dy = ops.h2_pack(x, xb), ops.h2_pack(dy, yb)
with ops.mfma_bf16(bf, bf):
    for _ in range(3):
        ops.conv2d(img(x), w, None, img(y), 3, 1)
        ops.conv2d_wgrad(img(x), img(dy), 3, 1, dw)
torch.cuda.synchronize()
PY
for BF in 0 1; do
  out=$O/${TAG}_pmc_conv_$([ $BF = 1 ] && echo bf16 || echo f32).txt
  rm -f $out
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    BF=$BF rocprofv3 --kernel-trace --pmc $set -d /tmp/pc${BF}_$i -o p -- python /tmp/one.py > /tmp/pc${BF}_$i.log 2>&1
    for k in conv_halo wgrad3_; do
      python $R/tools/rocpd_pmc.py --raw $k $(find /tmp/pc${BF}_$i -name "*.db" | head -1) >> $out 2>&1 || tail -3 /tmp/pc${BF}_$i.log >> $out
    done
  done
  BF=$BF rocprofv3 --kernel-trace --stats -d /tmp/pcs$BF -o p -- python /tmp/one.py > /dev/null 2>&1
  stats /tmp/pcs$BF | grep -E "conv_halo|wgrad3_" >> $out
done
head -8 $O/${TAG}_train_kernel_stats.txt; head -4 $O/${TAG}_bf16_train_kernel_stats.txt; cut -c1-400 $O/${TAG}_bench.json; tail -4 $O/${TAG}_pmc_conv_bf16.txt
