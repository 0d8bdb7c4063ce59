# early flush of the residual epilogues (specialised instantiations): old order vs row block by row block
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
export CHECK_KINDS=res_rowstats,res
{
run() { DIST_AMD_FAST_EARLY_FLUSH=$2 timeout 900 python tools/check_pp.py run $1 | grep -v "^$" > /dev/null; }
run f0 0; run f1 1; run f0b 0; run f1b 1
for t in f1 f0b f1b; do python tools/check_pp.py cmp f0 $t; done
} > gpurun_out/r05_ef2.log 2>&1
grep -E "RESULT|DIFF|SAME" gpurun_out/r05_ef2.log | grep -E "RESULT|DIFF|50432"
