cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_mixup.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 200 python tools/bench_mixup.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_mixup_bench.txt
