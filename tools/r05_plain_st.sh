# streaming hint of the GEMM's output stores on / off (timing-only library): time per launch alone and the step
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
export CHECK_KINDS=lnfold_act,lnfold_heads,res_rowstats
{
run() { DIST_AMD_FAST_PLAIN_ST=$2 timeout 900 python tools/check_pp.py run $1 | grep -v "^$" > /dev/null; }
run n0 0; run n1 1; run n0b 0; run n1b 1
for t in n1 n0b n1b; do python tools/check_pp.py cmp n0 $t; done
for v in 0 1 0 1; do DIST_AMD_FAST_PLAIN_ST=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain_st=$v ms/step', d['ms_per_step'])"; done
} > gpurun_out/r05_plain_st.log 2>&1
grep -E "RESULT|SAME|DIFF|plain_st" gpurun_out/r05_plain_st.log | grep -E "RESULT|50432|plain_st"
