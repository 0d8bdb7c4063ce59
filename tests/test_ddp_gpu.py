"""The multi-rank train step on the GPU: two ranks share the box's single GPU over gloo (RCCL refuses duplicate
devices), which exercises everything of the data-parallel path except RCCL itself: the engine's gradient-ready hook,
bucket coalescing, the side communication stream and its events, the 1/world scale folded into AdamW."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("gname,b,prec", [("tiny", 2, "fp32"), ("tiny", 2, "bf16")])
def test_two_ranks_one_gpu(gname, b, prec):
    env = dict(os.environ, DIST_AMD_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tools", "ddp_check.py"), gname, str(b), prec]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DDP_CHECK" in r.stdout and "-> OK" in r.stdout, r.stdout[-2000:]
