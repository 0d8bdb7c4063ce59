# where the frozen-ViT pass of batch n+1 sits inside step n: DIST_AMD_VIT_SPLIT layers before the branch forward, the rest ordered behind loss (ms per step)
cd /root/repo; mkdir -p gpurun_out
for i in 1 2; do
  for v in 12 9 6 3 0; do
    r=$(DIST_AMD_VIT_SPLIT=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])")
    echo "DIST_AMD_VIT_SPLIT=$v: $r ms"
  done
done
