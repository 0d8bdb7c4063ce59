# round 6, first call: baseline snapshot on this round's box, attention counters, forward-only sweeps (item 3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/snapshot.sh r06a
bash tools/r04_pmc_attn.sh > /dev/null 2>&1; mv gpurun_out/r04_pmc_attn.txt gpurun_out/r06_pmc_attn.txt
{
for rep in 1 2; do
  python tools/fwd_only.py; python tools/fwd_only.py --serial; python tools/fwd_only.py --vit-only; python tools/fwd_only.py --branch-only
done
. tools/measure_build.sh
for rep in 1 2; do for cfg in "1 1" "0 1" "2 1" "0 0" "2 0" "1 0"; do set -- $cfg
  DIST_AMD_PF_PRIO=$1 DIST_AMD_SIDE_PRIO=$2 python tools/fwd_only.py
done; done
} 2>&1 | grep -v "^$" | tee gpurun_out/r06_fwd_only_sweep.log
