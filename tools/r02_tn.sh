cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_tn" 2>&1 | tail -2
DIST_AMD_TN_MODES=0 timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_tn" 2>&1 | tail -1
out=gpurun_out/r02_tn_incremental.txt; : > $out
for v in 1 0 1 0; do
  echo "== DIST_AMD_TN_MODES=$v" >> $out
  DIST_AMD_TN_MODES=$v timeout 300 python tools/bench_cold.py 2>&1 | grep "gemm_tn" | grep -E "taps=[39]" >> $out
done
cat $out
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
bash tools/r02_step_ab.sh DIST_AMD_TN_MODES=0 DIST_AMD_TN_MODES=1
