# ping-pong GEMM vs gemm_fast8p: bit-identity (small shapes with an 8-block grid, full shapes) and time per launch
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh      # gemm_pp.hip only exists in the timing-only library
{
DIST_AMD_PP_GRID=8 DIST_AMD_FAST_PP=0 timeout 600 python tools/check_pp.py run s_old
DIST_AMD_PP_GRID=8 DIST_AMD_FAST_PP=1 timeout 600 python tools/check_pp.py run s_new
python tools/check_pp.py cmp s_old s_new
DIST_AMD_FAST_PP=0 timeout 900 python tools/check_pp.py run b_old
DIST_AMD_FAST_PP=1 timeout 900 python tools/check_pp.py run b_new
python tools/check_pp.py cmp b_old b_new
} > gpurun_out/r05_pp_check.log 2>&1
tail -60 gpurun_out/r05_pp_check.log
