# column-tile groups of the 256 x 256 GEMM beyond round 5's 1 ... 4 (timing-only library, DIST_AMD_FAST_NG forces the count for every shape; clamped to the column tiles): lone ViT pass
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
. tools/measure_build.sh
for rep in 1 2; do for ng in 0 2 3 5 6 9 12; do
  DIST_AMD_FAST_NG=$ng python tools/fwd_only.py --vit-only 2>&1 | grep fwd_only | sed "s/^/NG=$ng /"
done; done | tee gpurun_out/r06_ng.log
