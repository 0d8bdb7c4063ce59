# GEMM epilogue A/B in the timing-only library: early flush of the primary output (DIST_AMD_FAST_EARLY_FLUSH) and the epilogue's vectors
# prefetched into LDS by the prologue (DIST_AMD_FAST_AUX_LDS): bit-identity and time per launch against the old order
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
export CHECK_KINDS=plain,lnfold_act,lnfold_heads,res_rowstats
{
run() { DIST_AMD_FAST_EARLY_FLUSH=$2 DIST_AMD_FAST_AUX_LDS=$3 timeout 900 python tools/check_pp.py run $1 | grep -v "^$" > /dev/null; }
run e00 0 0; run e10 1 0; run e11 1 1; run e01 0 1; run e00b 0 0; run e10b 1 0; run e11b 1 1
for t in e10 e11 e01 e00b e10b e11b; do python tools/check_pp.py cmp e00 $t; done
} > gpurun_out/r05_ef_check.log 2>&1
grep -E "RESULT|DIFF|SAME" gpurun_out/r05_ef_check.log | grep -E "RESULT|DIFF|50432"
