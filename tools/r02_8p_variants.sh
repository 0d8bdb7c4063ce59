cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 2 4 6; do
  DIST_AMD_FAST_VAR=$v timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt_plain or gemm_fast" 2>&1 | tail -1
done
out=gpurun_out/r02_8p_variants.txt; : > $out
for cfg in "0 0" "1 0" "2 0" "4 0" "6 0" "0 2" "0 4" "0 6" "2 4" "0 0"; do
  set -- $cfg
  DIST_AMD_FAST_VAR=$1 DIST_AMD_FAST_STAGGER=$2 timeout 200 python tools/bench_fast8p.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
