# A/B of one selector in the pipelined step: bash tools/r05_step_ab.sh KNOB [reps]   (alternating runs on one box, ms per step)
cd /root/repo; mkdir -p gpurun_out
K=${1:-DIST_AMD_CONV9}; R=${2:-3}
for i in $(seq 1 $R); do
  for v in 0 1; do
    r=$(env $K=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])")
    echo "$K=$v: $r ms"
  done
done
