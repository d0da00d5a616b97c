# Timeline evidence for "gradient all-reduce overlapped with backward" on ONE GPU: the bench under torch.distributed.run with
# a 1-rank RCCL group and DF_FORCE_COLLECTIVES=1 (every bucketed all-reduce is really issued from inside the backward),
# kernel-traced; tools/rocpd_overlap.py then measures how much of the RCCL kernels' time ran beside compute kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
DF_FORCE_COLLECTIVES=1 rocprofv3 --kernel-trace -d /tmp/ov -o ov -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /tmp/ov.log 2>&1
tail -2 /tmp/ov.log | cut -c1-300
for db in $(find /tmp/ov -name "*.db"); do python $R/tools/rocpd_overlap.py $db; done > $R/gpurun_out/${TAG}_allreduce_overlap.txt 2>&1
cat $R/gpurun_out/${TAG}_allreduce_overlap.txt
