"""Measured parity gaps of the last GPU run -> gpurun_out/parity_gaps.json (so every gate in the GPU tests can be re-derived as ~2x measured)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEASURED = {}


def record(key, value):
    MEASURED[key] = float(value)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_gaps.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old.update(MEASURED)
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass
    return float(value)
