cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pg$i -o p -- python $R/tools/bench_gru.py > /tmp/pg$i.log 2>&1
  python $R/tools/rocpd_pmc.py --raw gru_bwd3 $(find /tmp/pg$i -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py --raw "gru_fwd3_kernel<true>" $(find /tmp/pg$i -name "*.db" | head -1) | head -6
done
