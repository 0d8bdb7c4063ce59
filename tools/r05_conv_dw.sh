cd /root/repo; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd /root/repo
{
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3_frame_resident or taps_conv_layout" 2>&1 | tail -3
DIST_AMD_CONV9=0 timeout 300 python tools/bench_conv_dw.py
timeout 300 python tools/bench_conv_dw.py
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c9 -o c9 -- python3 tools/bench_conv_dw.py > /dev/null 2>&1
python - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/prof_c9/c9_results.db")
for r in db.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, avg(grid_x/workgroup_x) from kernels group by name, grid_x order by name, grid_x"):
    if "conv3x3" in r[0]: print(f"{r[0][:70]:70s} n={r[1]:3d} avg {r[2]:7.1f} us  min {r[3]:7.1f}  max {r[4]:7.1f}  blocks {r[5]:.0f}")
PY
rm -rf gpurun_out/prof_c9
} > gpurun_out/r05_conv_dw.log 2>&1
cat gpurun_out/r05_conv_dw.log
