# usage: bash tools/pmc_pass2.sh NAME "COUNTER1 COUNTER2 ..."   (one --pmc set per pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 420 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d gpurun_out/pmc_$1 -o p -- python3 tools/pmc_fast.py > gpurun_out/pmc_$1.log 2>&1
echo rc=$? >> gpurun_out/pmc_$1.log
