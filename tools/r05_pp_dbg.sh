# timing-only ablation of the ping-pong GEMM (measure build): which part of a phase bounds it
cd /root/repo; mkdir -p gpurun_out
export DIST_AMD_BUILD_DEFS="-DPP_DEV_ONE"
. tools/measure_build.sh
{
for dbg in 42 106 234 170; do
  echo "=== DIST_AMD_PP_DBG=$dbg"
  DIST_AMD_FAST_PP=1 DIST_AMD_PP_DBG=$dbg timeout 300 python tools/bench_pp_plain.py
done
} > gpurun_out/r05_pp_dbg.log 2>&1
cat gpurun_out/r05_pp_dbg.log
