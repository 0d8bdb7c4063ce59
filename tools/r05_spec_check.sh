# straight-line instantiations of the ViT's three GEMM epilogues (DIST_AMD_FAST_SPEC) against the generic epilogue: bit-identity + time per launch
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
export CHECK_KINDS=plain,lnfold_act,lnfold_heads,res_rowstats
{
run() { DIST_AMD_FAST_SPEC=$2 timeout 900 python tools/check_pp.py run $1 | grep -v "^$" > /dev/null; }
run s0 0; run s1 1; run s0b 0; run s1b 1
for t in s1 s0b s1b; do python tools/check_pp.py cmp s0 $t; done
for d in 0 1 2; do DIST_AMD_FAST_DBG=$d timeout 600 python tools/bench_gemm_fixed.py dbg$d; done
} > gpurun_out/r05_spec_check.log 2>&1
grep -E "RESULT|DIFF|SAME|dbg" gpurun_out/r05_spec_check.log | grep -E "RESULT|DIFF|50432|dbg"
