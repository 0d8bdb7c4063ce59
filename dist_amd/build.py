"""Build libdist_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdist_amd.so")
SOURCES = ["gemm_nt.hip", "gemm_fast.hip", "gemm_pp.hip", "gemm_small.hip", "gemm_tn.hip", "gemm_tn8p.hip", "tnet.hip", "integ.hip", "norm.hip", "attn.hip", "misc.hip", "metrics.hip", "quant.hip", "engine.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "dist_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + os.environ.get("DIST_AMD_BUILD_DEFS", "").split() + ["-c", s, "-o", o]     # e.g. -DDIST_INTEG_ABLATE (tools/integ_ablate.sh)
            if verbose:
                print("[dist_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {src}")
    if procs or force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[dist_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
