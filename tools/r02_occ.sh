cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_nt_occ.txt; : > $out
for o in 0 5 0 5; do
  echo "== DIST_AMD_NT_OCC=$o" >> $out
  DIST_AMD_NT_OCC=$o timeout 300 python tools/bench_cold.py 2>&1 | grep "gemm_nt" | grep -v vit_ >> $out
done
for o in 0 1 5 0 1 5; do
  DIST_AMD_NT_OCC=$o timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ=$o ms/step', d['ms_per_step'])" >> $out
done
cat $out
