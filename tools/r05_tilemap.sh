# the tile map's divisions on the host (DIST_AMD_FAST_TILEMAP) against divisions in the kernel: bit-identity + time per launch (timing-only library)
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
export CHECK_KINDS=plain,lnfold_act,lnfold_heads,res_rowstats
{
run() { DIST_AMD_FAST_TILEMAP=$2 timeout 900 python tools/check_pp.py run $1 | grep -v "^$" > /dev/null; }
run t0 0; run t1 1; run t0b 0; run t1b 1
for t in t1 t0b t1b; do python tools/check_pp.py cmp t0 $t; done
} > gpurun_out/r05_tilemap.log 2>&1
grep -E "RESULT|DIFF|SAME" gpurun_out/r05_tilemap.log | grep -E "RESULT|DIFF|50432|65792|20037"
