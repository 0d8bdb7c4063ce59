# whole-suite check + step bench (pipelined and serial order) + phase times
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-x}
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r02_${tag}_tests.txt
cat gpurun_out/r02_${tag}_tests.txt
timeout 300 python tools/phase_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_${tag}_phase_times.txt
cat gpurun_out/r02_${tag}_phase_times.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_${tag}_bench.json 2> gpurun_out/r02_${tag}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_${tag}_bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "clips/s", d["value"], "serial", d.get("serial_order",{}).get("ms_per_step"))
print("roofline", {k:d["roofline"][k] for k in ("achieved","frac","avg_launch_us","launches_per_step")}, "alone", d["roofline"].get("alone"))
PY
