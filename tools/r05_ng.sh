# column-tile groups of the 256 x 256 kernel (DIST_AMD_FAST_NG, timing-only library) on the ViT shapes with today's epilogues
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
export CHECK_KINDS=lnfold_act,lnfold_heads,res_rowstats
{
for ng in 0 1 2 3 4; do DIST_AMD_FAST_NG=$ng timeout 900 python tools/check_pp.py run g$ng | grep -v "^$" > /dev/null; done
for ng in 1 2 3 4; do python tools/check_pp.py cmp g0 g$ng; done
} > gpurun_out/r05_ng.log 2>&1
grep -E "SAME|DIFF" gpurun_out/r05_ng.log | grep -E "50432x2304x768:lnfold_heads|50432x3072x768:lnfold_act|50432x768x3072:res_rowstats|50432x768x768:res_rowstats"
