"""The RCCL leg of the data-parallel path, executed on the box the tests run on (one GPU): `bench.py` with backend `nccl` (= RCCL on
ROCm) and the gradient reducer forced on at world size 1 (DIST_AMD_FORCE_REDUCER=1).  Everything the N-GPU run uses is exercised except
the wire: `init_process_group(backend="nccl", device_id=...)`, the engine's gradient-ready hook, bucket coalescing, the communication
stream and its events, one `ncclAllReduce` per bucket over the flat dist_net gradient buffer (every element exactly once), the 1/world
scale in AdamW - and the step must not slow down: a HIGH-priority stream beside the engine's four once cost 8-15 ms per step
(DESIGN.md section 5), which is exactly what RCCL's own streams could reproduce.  (Two ranks on one GPU need gloo - RCCL refuses duplicate
devices - and are covered by tests/test_ddp_gpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(env_extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", **env_extra)
    env.pop("DIST_AMD_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline", "--no-serial-ref"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_rccl_reducer_path_at_world_size_one():
    plain = run_bench({})
    forced = run_bench({"DIST_AMD_FORCE_REDUCER": "1"})
    plain2 = run_bench({})
    red = forced["reducer"]
    assert "reducer" not in plain
    assert red["backend"] == "nccl" and red["world"] == 1 and red["forced_at_world_1"] and red["overlap"]
    assert red["collectives_per_step"] >= 2                                   # several buckets: the all-reduce of the last layers overlaps the rest
    assert red["elements_reduced_per_step"] == red["grad_elements"] == 19001184   # every dist_net gradient element exactly once (76 MB fp32)
    base = min(plain["ms_per_step"], plain2["ms_per_step"])
    print(f"step without reducer {plain['ms_per_step']:.3f} / {plain2['ms_per_step']:.3f} ms, with RCCL reducer at world 1 {forced['ms_per_step']:.3f} ms, "
          f"{red['collectives_per_step']} all-reduces of <= {red['bucket_bytes'] >> 20} MB per step")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _gaps import record
    record("rccl.world1.step_ratio", forced["ms_per_step"] / base)
    assert forced["ms_per_step"] <= 1.05 * base, (forced["ms_per_step"], base)
