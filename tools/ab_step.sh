#!/bin/bash
# A/B of one environment knob on the pipelined train step, alternating on ONE box: tools/ab_step.sh VAR A B [pairs]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for v in $A $B; do
    r=$(env $VAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('forward_only',{}).get('ms_per_iteration'))")
    echo "$VAR=$v: $r"
  done
done
