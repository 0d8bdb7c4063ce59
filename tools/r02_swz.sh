cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt or conv3x3 or outmaps or rowmaps" 2>&1 | tail -2
out=gpurun_out/r02_nt_swz.txt; : > $out
for o in 1 0 1 0; do
  echo "== DIST_AMD_NT_OCC=$o" >> $out
  DIST_AMD_NT_OCC=$o timeout 300 python tools/bench_cold.py 2>&1 | grep "gemm_nt" | grep -v vit_ >> $out
done
cat $out
rm -rf gpurun_out/pmc_nt
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_nt -o p -- python3 tools/pmc_nt_shapes.py > gpurun_out/pmc_nt.log 2>&1
python3 tools/pmc_generic.py gpurun_out/pmc_nt gemm_nt
rm -rf gpurun_out/pmc_nt
