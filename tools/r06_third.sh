cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_dropin_gpu.py -x -q -k train_loop_pipelines 2>&1 | tail -40 > gpurun_out/r06_dropin.log
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --host-input --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host-input on the prefetch stream', d['ms_per_step'])"
DIST_AMD_HOST_INPUT=own python bench.py --steps 20 --warmup 5 --host-input --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('host-input on its own copy stream', d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', d['ms_per_step'])"
done | tee gpurun_out/r06_host_input.log
