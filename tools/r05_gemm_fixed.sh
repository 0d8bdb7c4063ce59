# where the fixed part of a K = 768 GEMM tile goes (timing-only library)
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
{
for d in 0 1 2 0; do DIST_AMD_FAST_DBG=$d timeout 600 python tools/bench_gemm_fixed.py dbg$d; done
} > gpurun_out/r05_gemm_fixed.log 2>&1
cat gpurun_out/r05_gemm_fixed.log | grep -v "^$" | tail -20
