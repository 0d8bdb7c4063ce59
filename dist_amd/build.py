"""Build libdist_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdist_amd.so")
MEASURE_LIB = os.path.join(CSRC, "libdist_amd_measure.so")
SOURCES = ["gemm_nt.hip", "gemm_fast.hip", "gemm_small.hip", "gemm_tn.hip", "gemm_tn8p.hip", "conv_dw.hip", "conv_t_dw.hip", "tnet.hip", "integ.hip", "norm.hip", "attn.hip", "misc.hip",
           "metrics.hip", "quant.hip", "engine.hip", "engine_vit.hip", "engine_fwd.hip", "engine_bwd.hip"]
# kernels of the timing-only library alone (measured-and-rejected variants kept as A/B references): never compiled into libdist_amd.so
MEASURE_SOURCES = [os.path.join("measure", "gemm_pp.hip"), os.path.join("measure", "integ4.hip")]
HEADERS = ["common.h", "kernels.h", "engine_internal.h", "integ_common.h", os.path.join("..", "..", "include", "dist_amd.h"),
           os.path.join("measure", "gemm_fast8q.inl"), os.path.join("measure", "gemm_fast8q_launch.inl")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=True, measure=False):
    """Product build: csrc/*.o -> csrc/libdist_amd.so.  measure=True: the timing-only build (-DDIST_AMD_MEASURE plus DIST_AMD_BUILD_DEFS: kernels
    can be skipped, dummy work, the rejected A/B variants) goes to ITS OWN objects (*.m.o) and library (libdist_amd_measure.so, selected with
    DIST_AMD_LIB=...), so a measurement can never leave a wrong-result library where the product one is expected.  The extra definitions are part of
    the staleness key of either build (csrc/.defs[.measure]): changing them rebuilds everything."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    defs = os.environ.get("DIST_AMD_BUILD_DEFS", "").split()
    if measure and "-DDIST_AMD_MEASURE" not in defs:
        defs = ["-DDIST_AMD_MEASURE"] + defs
    if not measure and "-DDIST_AMD_MEASURE" in defs:
        raise RuntimeError("a -DDIST_AMD_MEASURE build is `python -m dist_amd.build --measure` (its own library), not the product build")
    lib = MEASURE_LIB if measure else LIB
    osuf = ".m.o" if measure else ".o"
    key_file = os.path.join(CSRC, ".defs.measure" if measure else ".defs")
    key = " ".join(defs)
    if (open(key_file).read() if os.path.exists(key_file) else "") != key:
        force = True
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    sources = SOURCES + (MEASURE_SOURCES if measure else [])
    srcs = [os.path.join(CSRC, x) for x in sources if os.path.exists(os.path.join(CSRC, x))]
    if not force and not _stale(lib, srcs + hdrs):
        return lib                                       # (a snapshot on the GPU box carries the library but not the objects)
    objs = []
    procs = []
    for src in sources:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace(".hip", osuf))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + defs + ["-c", s, "-o", o]
            if verbose:
                print("[dist_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed on {src}")
    if procs or force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print("[dist_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(key_file, "w") as f:
        f.write(key)
    return lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, measure="--measure" in sys.argv)
