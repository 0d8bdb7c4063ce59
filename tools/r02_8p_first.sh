cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt_plain or gemm_fast or heads_outmap or folded" > gpurun_out/r02_8p_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/r02_8p_tests.txt
tail -5 gpurun_out/r02_8p_tests.txt
DIST_AMD_FAST_8P=0 timeout 200 python tools/bench_fast8p.py > gpurun_out/r02_8p_ab.txt 2>&1
timeout 200 python tools/bench_fast8p.py >> gpurun_out/r02_8p_ab.txt 2>&1
DIST_AMD_FAST_8P=0 timeout 200 python tools/bench_fast8p.py >> gpurun_out/r02_8p_ab.txt 2>&1
timeout 200 python tools/bench_fast8p.py >> gpurun_out/r02_8p_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r02_8p_ab.txt
