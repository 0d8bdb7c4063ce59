# the multi-tile GEMM blocks (measure library) in the pipelined step: ms per step, alternating
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
for i in 1 2; do
  for v in 0 32 8 4; do
    r=$(DIST_AMD_FAST_TILES=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['measure_build'])")
    echo "DIST_AMD_FAST_TILES=$v: $r"
  done
done
