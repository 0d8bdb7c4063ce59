# counters of the fused IntegrationNetwork kernels, 8 waves / 128 rows (DIST_AMD_INTEG_W4=0) against 4 waves / 64 rows x 2 workgroups per CU
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r06_pmc_integ4.txt; : > $out
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC)_[A-Z0-9_a-z]+" | sort -u | tr '\n' ' ' > gpurun_out/r06_counters_tcp.txt
for w4 in 0 1; do
  export DIST_AMD_INTEG_W4=$w4
  echo "===== DIST_AMD_INTEG_W4=$w4" >> $out
  rm -rf gpurun_out/kt_ig
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_ig -o p -- python3 tools/pmc_integ4_shape.py > gpurun_out/kt_ig.log 2>&1
  f=$(find gpurun_out/kt_ig -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "integ_" in r["Name"] and "pack" not in r["Name"]]
for r in rows:
    print(f"{r['Calls']:>5} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:70]}")
PY
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT" \
             "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
             "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
    rm -rf gpurun_out/pmc_ig
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_ig -o p -- python3 tools/pmc_integ4_shape.py > gpurun_out/pmc_ig.log 2>&1
    grep -i "error\|invalid\|not found" gpurun_out/pmc_ig.log | head -2 >> $out
    python3 tools/pmc_generic.py gpurun_out/pmc_ig integ_ 2>&1 | grep -v "integ_pack" >> $out
  done
done
rm -rf gpurun_out/pmc_ig gpurun_out/kt_ig
