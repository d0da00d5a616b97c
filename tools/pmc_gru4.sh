# SQ / LDS / L2 counters of the lean GRU kernels inside the training step (one rocprofv3 --pmc pass per counter set; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pg$i -o p -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-loader --no-profile > /tmp/pg$i.log 2>&1
  db=$(find /tmp/pg$i -name "*.db" | head -1)
  if [ -z "$db" ]; then echo "pass $i ($set): no db"; tail -3 /tmp/pg$i.log; continue; fi
  for k in ${PMC_KERNELS:-gru_fwd4 gru_bwd4 gru_wgrad4}; do python $R/tools/rocpd_pmc.py --raw $k $db; done
done
