# column tiles of a row panel started apart (timing-only library, DIST_AMD_FAST_SKEW = units of 64 cycles per column index, first round only): lone ViT pass and step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
. tools/measure_build.sh
for rep in 1 2; do for sk in 0 2 4 8 16 32; do
  DIST_AMD_FAST_SKEW=$sk python tools/fwd_only.py --vit-only 2>&1 | grep fwd_only | sed "s/^/SKEW=$sk /"
done; done | tee gpurun_out/r06_skew.log
for sk in 0 4 8 16 0; do DIST_AMD_FAST_SKEW=$sk python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SKEW=$sk step', d['ms_per_step'])"; done | tee -a gpurun_out/r06_skew.log
