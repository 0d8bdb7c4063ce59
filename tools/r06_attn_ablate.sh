# attention kernel: what its phases cost (timing-only library: DIST_AMD_ATTN_DBG 1 = staging only, 2 = no K / V staging loads, 3 = neither)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
. tools/measure_build.sh
for d in 0 1 2 3 0; do DIST_AMD_ATTN_DBG=$d python tools/bench_attn.py 2>&1 | grep "B/16" | sed "s/^/DBG=$d /"; done | tee gpurun_out/r06_attn_ablate.log
