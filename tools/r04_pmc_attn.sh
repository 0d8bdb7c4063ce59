# kernel trace + PMC passes of the ViT-B/16 attention kernel alone
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_pmc_attn.txt; : > $out
rm -rf gpurun_out/kt_at
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_at -o p -- python3 tools/pmc_attn_shape.py > gpurun_out/kt_at.log 2>&1
f=$(find gpurun_out/kt_at -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:3]:
    print(f"{r['Calls']:>5} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:90]}")
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  rm -rf gpurun_out/pmc_at
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_at -o p -- python3 tools/pmc_attn_shape.py > gpurun_out/pmc_at.log 2>&1
  tail -1 gpurun_out/pmc_at.log >> $out
  python3 tools/pmc_generic.py gpurun_out/pmc_at attn_kernel >> $out 2>&1
done
rm -rf gpurun_out/pmc_at gpurun_out/kt_at
cat $out
