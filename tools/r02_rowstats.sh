cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "row_statistics or folded or gemm_fast or gemm_nt_plain" 2>&1 | tail -3
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/r02_step_ab.sh DIST_AMD_ROWSTATS=0 DIST_AMD_ROWSTATS=1
for v in 0 1; do DIST_AMD_ROWSTATS=$v timeout 200 python tools/vit_alone.py 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/[rowstats=$v] /"; done
