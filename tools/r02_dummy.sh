cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_dummy_stores.txt; : > $out
for d in 0 1 0 1; do
  DIST_AMD_FAST_DUMMY=$d timeout 200 python tools/bench_fast8p.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[dummy=$d] /" >> $out
done
cat $out
