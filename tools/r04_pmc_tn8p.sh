# kernel trace + PMC passes of the two-group weight-gradient kernel (gemm_tn8p.hip) on the input_linear gradient: bash tools/r04_pmc_tn8p.sh [NI K]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NI=${1:-384}; K=${2:-768}
out=gpurun_out/r04_pmc_tn8p_${NI}x${K}.txt; : > $out
rm -rf gpurun_out/kt_tn
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tn -o p -- python3 tools/pmc_tn_shape.py $NI $K > gpurun_out/kt_tn.log 2>&1
f=$(find gpurun_out/kt_tn -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f"{r['Calls']:>5} calls  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:90]}")
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  rm -rf gpurun_out/pmc_tn
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_tn -o p -- python3 tools/pmc_tn_shape.py $NI $K > gpurun_out/pmc_tn.log 2>&1
  tail -1 gpurun_out/pmc_tn.log >> $out
  python3 tools/pmc_generic.py gpurun_out/pmc_tn tn8 >> $out 2>&1
done
rm -rf gpurun_out/pmc_tn gpurun_out/kt_tn
cat $out
