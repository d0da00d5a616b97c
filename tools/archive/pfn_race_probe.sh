#!/bin/bash
# Round 4: what does the pillar feature net's backward need in order to return wrong sums?
#   lib: default (pillarize.hip built with -fno-slp-vectorize)  |  slp (the same sources, SLP vectoriser on: packed-fp32 VALU)
#   neighbour: none | a bf16-MFMA side STREAM in the same process | a second PROCESS running the bf16 training step
# usage (GPU box): bash tools/pfn_race_probe.sh > gpurun_out/pfn_race_probe.txt
cd ${GRAFT_REPO_ROOT:-.}
SLP=$PWD/deflow_amd/_build/deflow_amd_slp/libdeflow_amd_slp.so
REPS=${REPS:-20000}
run() { "$@" 2>&1 | grep -E "pfn backward|reps differ" ; }
echo "== slp build, alone";                         DF_LIB=$SLP run python tools/pfn_bwd_stress.py $REPS
echo "== slp build, bf16 GEMM side stream, same process";  DF_LIB=$SLP DF_STRESS_SIDE=matmul run python tools/pfn_bwd_stress.py $REPS
echo "== slp build, second process (bf16 training step, default build)"
python tools/grad_repro_probe.py 1000000 bf16 > /tmp/nb1.log 2>&1 &
NB=$!
sleep 20
DF_LIB=$SLP run python tools/pfn_bwd_stress.py $REPS
kill $NB 2>/dev/null; wait $NB 2>/dev/null
echo "== default build, second process"
python tools/grad_repro_probe.py 1000000 bf16 > /tmp/nb2.log 2>&1 &
NB=$!
sleep 20
run python tools/pfn_bwd_stress.py $REPS
kill $NB 2>/dev/null; wait $NB 2>/dev/null
echo "== slp build, second process running bf16 GEMMs only (torch / hipBLASLt kernels, none of this library's)"
python - > /tmp/nb3.log 2>&1 <<'PY' &
import torch, time
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); b = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 90:
    for _ in range(200): c = a @ b
    torch.cuda.synchronize()
PY
NB=$!
sleep 15
DF_LIB=$SLP run python tools/pfn_bwd_stress.py $REPS
kill $NB 2>/dev/null; wait $NB 2>/dev/null
