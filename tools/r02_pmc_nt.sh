cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_pmc_nt.txt; : > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum"; do
  rm -rf gpurun_out/pmc_nt
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_nt -o p -- python3 tools/pmc_nt_shapes.py > gpurun_out/pmc_nt.log 2>&1
  tail -2 gpurun_out/pmc_nt.log >> $out
  python3 tools/pmc_generic.py gpurun_out/pmc_nt gemm_nt >> $out 2>&1
done
rm -rf gpurun_out/pmc_nt
cat $out
