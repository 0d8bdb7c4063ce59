# multi-tile GEMM blocks (gemm_fast8q_kernel, DIST_AMD_FAST_TILES) vs one tile per block (gemm_fast8p_kernel): bit-identity and time per launch
cd /root/repo; mkdir -p gpurun_out
export CHECK_KINDS=plain,lnfold_act,lnfold_heads
{
for t in 0 32 8 4 0b 32b 8b 4b; do DIST_AMD_FAST_TILES=${t%b} timeout 900 python tools/check_pp.py run q$t | grep -v "^$"; done
for t in 32 8 4 0b 32b 8b 4b; do python tools/check_pp.py cmp q0 q$t; done
} > gpurun_out/r05_q_check.log 2>&1
grep -E "RESULT|DIFF|SAME" gpurun_out/r05_q_check.log | grep -E "RESULT|DIFF|50432|65792x4096" 
