# counters of the dominant GEMM on the four ViT shapes (tools/pmc_fast.py), separate passes per set: where the K loop waits
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r06_pmc_gemm.txt; : > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  rm -rf gpurun_out/pmc_g
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_g -o p -- python3 tools/pmc_fast.py > gpurun_out/pmc_g.log 2>&1
  python3 tools/pmc_generic.py gpurun_out/pmc_g gemm_fast8p >> $out 2>&1
done
rm -rf gpurun_out/pmc_g
