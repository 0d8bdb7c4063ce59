# the other BASELINE configurations through bench.py on one MI355X (parity for them: tests/test_configs_gpu.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "b16_16+32f 32 0" "l14_32+64f 8 0" "l14_32+64f 16 31"; do
  set -- $cfg
  timeout 600 python bench.py --config $1 --batch $2 --vit-fp8 $3 --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-serial-ref 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 b=$2 fp8=$3', 'ms/step', d['ms_per_step'], 'clips/s', d['value'], 'path_mfma_frac', d.get('path_mfma_frac'))"
done
done 2>&1 | tee gpurun_out/r05_configs.log
