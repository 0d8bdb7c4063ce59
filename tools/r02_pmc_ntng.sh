cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_pmc_nt_ng.txt; : > $out
for cfg in "0 0" "1 0" "0 1" "1 1" "0 3" "2 0"; do
  set -- $cfg
  rm -rf gpurun_out/pmc_tmp
  DIST_AMD_FAST_NT=$1 DIST_AMD_FAST_NG=$2 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 tools/pmc_fast.py > gpurun_out/pmc_tmp.log 2>&1
  echo "== FETCH_SIZE x2, nt=$1 ng=$2 (order of shapes by blocks: 1773 qkv, 591 out/proj, 2364 fc)" >> $out
  python tools/pmc_csv.py gpurun_out/pmc_tmp FETCH_SIZE 2 >> $out 2>&1
done
rm -rf gpurun_out/pmc_tmp
cat $out
