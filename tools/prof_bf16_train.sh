cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/pb -o p -- python $R/tools/bench_bf16_train.py bf16 4 > /tmp/pb.log 2>&1
tail -1 /tmp/pb.log
python $R/tools/rocpd_stats.py $(find /tmp/pb -name "*.db" | head -1) | head -40
