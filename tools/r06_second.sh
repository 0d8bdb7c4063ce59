# round 6, second batch of measurements: two-half ViT experiment, forward budget, host-input bench + parity, the new bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/vit_two_halves.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_vit_two_halves.log
HALF_PRIO=-1 python tools/vit_two_halves.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/r06_vit_two_halves.log
bash tools/r06_fwd_budget.sh
python -m pytest tests/test_dropin_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06_dropin.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06b.json 2> gpurun_out/bench_r06b.err; tail -c 1500 gpurun_out/bench_r06b.json
python bench.py --steps 20 --warmup 5 --host-input --no-cpu-baseline > gpurun_out/bench_r06b_host.json 2>> gpurun_out/bench_r06b.err; tail -c 600 gpurun_out/bench_r06b_host.json
