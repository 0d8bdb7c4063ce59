# generic A/B of environment settings on the pipelined step, alternating on one box: bash tools/r04_step_ab.sh "VAR=a" "VAR=b" ...
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_step_ab.txt; : > $out
for i in 1 2 3; do
  for kv in "$@"; do
    r=$(env $kv python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$kv : $r" >> $out
  done
done
cat $out
