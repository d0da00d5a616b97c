cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 > /tmp/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) > $R/gpurun_out/r01_g_kernel_stats.txt 2>&1
tail -1 /tmp/kt.log > $R/gpurun_out/r01_g_bench_under_rocprof.json
head -50 $R/gpurun_out/r01_g_kernel_stats.txt
