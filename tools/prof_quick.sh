# kernel-trace stats of the training launches only (short): gpurun_out/<TAG>_train_kernel_stats.txt
TAG=${1:-quick}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_q
DF_BENCH_NO_SUBPROC=1 rocprofv3 --kernel-trace --stats -d /tmp/kt_q -o kt -- python $R/bench.py --no-extras --no-cpu-baseline --no-loader --steps ${STEPS:-5} > /tmp/kt_q.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt_q -name "*.db" | head -1) > $R/gpurun_out/${TAG}_train_kernel_stats.txt 2>&1
grep "^{\"metric\"" /tmp/kt_q.log | tail -1 > $R/gpurun_out/${TAG}_train_bench.json
head -${HEADN:-70} $R/gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-150
