# kernel trace of the fp32 training step (the headline):  gpurun -- bash tools/prof_fp32_train.sh <tag>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r03}
mkdir -p $R/gpurun_out; rm -rf /tmp/pf
rocprofv3 --kernel-trace -d /tmp/pf -o p -- python $R/tools/bench_bf16_train.py fp32 4 > /tmp/pf.log 2>&1
( tail -1 /tmp/pf.log; python $R/tools/rocpd_stats.py $(find /tmp/pf -name "*.db" | head -1) ) > $R/gpurun_out/${TAG}_fp32_train_kernel_stats.txt
head -42 $R/gpurun_out/${TAG}_fp32_train_kernel_stats.txt
