# usage: bash tools/pmc_pass.sh COUNTER   (one counter set per pass; no trace domains besides --kernel-trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 420 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d gpurun_out/pmc_$1 -o p -- python3 tools/pmc_fast.py > gpurun_out/pmc_$1.log 2>&1
echo rc=$? >> gpurun_out/pmc_$1.log
