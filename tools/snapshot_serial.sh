cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export DIST_AMD_SERIAL=3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o s -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof_s.log 2>&1
