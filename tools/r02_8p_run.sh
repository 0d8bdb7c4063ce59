cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt_plain or gemm_fast or heads_outmap or folded" 2>&1 | tail -3
out=gpurun_out/r02_8p_readahead.txt; : > $out
timeout 200 python tools/bench_fast8p.py 2>&1 | grep -v amdgpu.ids >> $out
DIST_AMD_FAST_8P=0 timeout 200 python tools/bench_fast8p.py 2>&1 | grep -v amdgpu.ids >> $out
timeout 200 python tools/bench_fast8p.py 2>&1 | grep -v amdgpu.ids >> $out
cat $out
