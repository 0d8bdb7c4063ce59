cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_pmc_tn.txt; : > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum"; do
  rm -rf gpurun_out/pmc_tn
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_tn -o p -- python3 tools/pmc_tn_shape.py 384 768 > gpurun_out/pmc_tn.log 2>&1
  tail -2 gpurun_out/pmc_tn.log >> $out
  python3 tools/pmc_generic.py gpurun_out/pmc_tn gemm_tn >> $out 2>&1
done
rm -rf gpurun_out/pmc_tn
cat $out
