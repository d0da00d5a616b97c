# round 6, second session: baseline of the rebuilt library on one box -- training bench, B = 1 forward per-layer table + kernel trace,
# pillar stage alone.  Results under gpurun_out/r06b_*
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r06b_bench_noextras.json 2> gpurun_out/r06b_bench_noextras.err
python tools/layer_table_infer.py 1 > gpurun_out/r06b_layer_infer_b1.txt 2>&1
python tools/bench_pillar.py 16 > gpurun_out/r06b_pillar.txt 2>&1
python tools/bench_pillar.py 1 >> gpurun_out/r06b_pillar.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_i
rocprofv3 --kernel-trace --stats -d /tmp/kt_i -o kt -- python $R/tools/prof_infer.py 1 > /tmp/kt_i.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/kt_i -name "*.db" | head -1) > $R/gpurun_out/r06b_infer_b1_kernel_stats.txt 2>&1
tail -2 /tmp/kt_i.log
cd $R
python tools/show_bench.py gpurun_out/r06b_bench_noextras.json | head -30
cat gpurun_out/r06b_pillar.txt
