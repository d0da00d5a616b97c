# kernel trace of the bf16 training step (serialised: DF_SIDE_STREAM=0, so per-kernel durations do not overlap), with and
# without bf16 storage:  gpurun -- bash tools/prof_bf16_train.sh <tag>
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r03}
mkdir -p $R/gpurun_out
for mode in store nostore; do
  rm -rf /tmp/pb
  if [ $mode = nostore ]; then export DF_BF16_STORE=0; else unset DF_BF16_STORE; fi
  DF_SIDE_STREAM=0 rocprofv3 --kernel-trace -d /tmp/pb -o p -- python $R/tools/bench_bf16_train.py bf16 4 > /tmp/pb.log 2>&1
  ( tail -1 /tmp/pb.log; python $R/tools/rocpd_stats.py $(find /tmp/pb -name "*.db" | head -1) ) > $R/gpurun_out/${TAG}_bf16_${mode}_serial_kernel_stats.txt
done
unset DF_BF16_STORE
head -45 $R/gpurun_out/${TAG}_bf16_store_serial_kernel_stats.txt
