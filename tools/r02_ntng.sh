cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_nt_ng.txt; : > $out
for cfg in "0 0" "1 0" "2 0" "3 0" "0 1" "1 1" "0 3" "1 3" "0 4" "1 2" "0 0"; do
  set -- $cfg
  DIST_AMD_FAST_NT=$1 DIST_AMD_FAST_NG=$2 timeout 200 python tools/bench_fast8p.py 2>&1 | grep -E "fc |qkv|proj|out " | sed "s/^.*stagger=0\]/[nt=$1 ng=$2]/" >> $out
done
cat $out
