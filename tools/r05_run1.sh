cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_engine_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "loss or import or column_offset or two_group or lnfold or fold or fast" > gpurun_out/r05_run1_tests.log 2>&1
tail -5 gpurun_out/r05_run1_tests.log
bash tools/snapshot.sh r05a
cat gpurun_out/bench_r05a.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k: d[k] for k in ('ms_per_step','value','path_mfma_frac','cu_time_floor_ms') if k in d}); print(d['roofline']['alone'], d['roofline']['frac'], d['roofline'].get('frac_profile')); print(d['forward_only'])"
head -40 gpurun_out/prof_r05a_summary.md
