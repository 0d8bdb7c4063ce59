# A/B of the whole step under environment knobs: bash tools/r02_step_ab.sh "VAR=a" "VAR=b" ...   (3 alternations)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_step_ab.txt; : > $out
for rep in 1 2 3; do
  for cfg in "$@"; do
    env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms/step', d['ms_per_step'])" >> $out
  done
done
sort $out | awk '{k=$1; s[k]+=$NF; n[k]++; a[k]=a[k]" "$NF} END {for (k in s) printf "%-40s mean %.3f  (%s )\n", k, s[k]/n[k], a[k]}'
