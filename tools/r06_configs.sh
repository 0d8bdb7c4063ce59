# BASELINE configurations 3 / 4 / 5 through bench.py with their kernel statistics (profiles/r06_cfg{3,4,5}_*): one MI355X
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
SNAP_STEPS=8 SNAP_WARMUP=3 SNAP_SERIAL_STEPS=4 bash tools/snapshot.sh cfg3 --config b16_16+32f --batch 32
SNAP_STEPS=8 SNAP_WARMUP=3 SNAP_SERIAL_STEPS=4 bash tools/snapshot.sh cfg4 --config l14_32+64f --batch 8
SNAP_STEPS=8 SNAP_WARMUP=3 SNAP_SERIAL_STEPS=4 bash tools/snapshot.sh cfg5 --config l14_32+64f --batch 16 --vit-fp8 31
for c in cfg3 cfg4 cfg5; do python -c "
import json; d=json.loads(open('gpurun_out/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['config']['workload'], 'ms/step', d['ms_per_step'], 'clips/s', d['value'], 'path_mfma_frac', d.get('path_mfma_frac'), d.get('path_peak', {}).get('peak_tflops'))"; done | tee gpurun_out/r06_configs.log
