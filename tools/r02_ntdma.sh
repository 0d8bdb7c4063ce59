cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt or conv3x3 or outmaps or rowmaps" 2>&1 | tail -3
out=gpurun_out/r02_nt_dma.txt; : > $out
for v in 1 0 1 0; do
  echo "== DIST_AMD_NT_DMA=$v" >> $out
  DIST_AMD_NT_DMA=$v timeout 300 python tools/bench_cold.py 2>&1 | grep "gemm_nt" | grep -v vit_ >> $out
done
cat $out
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
bash tools/r02_step_ab.sh DIST_AMD_NT_DMA=0 DIST_AMD_NT_DMA=1
