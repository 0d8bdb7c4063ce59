cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt or conv3x3 or outmaps or rowmaps" 2>&1 | tail -2
for r in 2 0; do echo "== DIST_AMD_NT_ROTATE=$r (2: no tile prefetch)"; DIST_AMD_NT_ROTATE=$r timeout 300 python tools/bench_cold.py 2>&1 | grep "gemm_nt" | grep -v vit_; done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for r in 2 0 2 0; do DIST_AMD_NT_ROTATE=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rotate $r ms/step', d['ms_per_step'], 'clips/s', d['value'])"; done
