# phase shares of the fused TemporalNet kernel (DIST_AMD_TNET_DBG bits; results are wrong with any bit set)
mkdir -p gpurun_out
. tools/measure_build.sh      # the debug switches only exist in the timing-only library
for d in 0 15 31; do echo "dbg=$d"; DIST_AMD_TNET_DBG=$d python tools/bench_tnet.py 2>&1 | grep "config 2" | grep "fused"; done
