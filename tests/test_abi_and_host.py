"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads and exports every
symbol include/dist_amd.h declares (no compute calls); host-side logic (synthetic data, engine-free
helpers, distributed wrappers over gloo with world_size 2)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    from dist_amd import build, lib
    path = build.build_library(verbose=False)
    assert os.path.exists(path)
    l = lib.load()
    syms = lib.exported_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(l, s), s
    assert l.dist_abi_version() == lib.ABI_VERSION == 9
    hdr = open(os.path.join(ROOT, "include", "dist_amd.h")).read()
    assert "#define DIST_ABI_VERSION 9" in hdr
    # the ctypes mirrors of the argument structs have the library's layout size (checked without a GPU)
    for cname, mirror in (("dist_gemm_args", lib.GemmArgs), ("dist_gemm_tn_args", lib.GemmTnArgs), ("dist_ln_args", lib.LnArgs),
                          ("dist_ln_bwd_args", lib.LnBwdArgs), ("dist_adamw_seg", lib.AdamwSeg), ("dist_config", lib.Config),
                          ("dist_rowmap", lib.RowMap), ("dist_outmap", lib.OutMap)):
        assert l.dist_abi_sizeof(cname.encode()) == ctypes.sizeof(mirror), cname
    assert l.dist_abi_sizeof(b"nope") == -1
    assert l.dist_strerror(-1).decode().startswith("invalid argument")


def test_integration_md_stub_matches_the_library():
    """The ctypes stub INTEGRATION.md shows a maintainer is extracted from the document and checked against the built library:
    its dist_config mirror has the library's size, its example initializer fills every field, and dist_create accepts it
    (host-only).  Round 2's stub had 17 of 18 fields: dist_create read 4 bytes past the caller's struct."""
    import re
    from dist_amd import lib
    l = lib.load()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(import ctypes as C, torch.*?)```", md, re.S).group(1)
    cls_src = re.search(r"(class dist_config\(C\.Structure\):\n(?:    .*\n|\s*\n)+?)\n", block).group(1)
    init_src = re.search(r"^cfg = (dist_config\(.*?\))", block, re.M).group(1)
    ns = {"C": ctypes}
    exec(cls_src, ns)
    stub = ns["dist_config"]
    assert ctypes.sizeof(stub) == l.dist_abi_sizeof(b"dist_config") == ctypes.sizeof(lib.Config)
    assert [n for n, _ in stub._fields_] == [n for n, _ in lib.Config._fields_]
    nargs = len(re.findall(r"-?\d+", init_src.split("(", 1)[1]))
    assert nargs == len(stub._fields_), "the example initializer must fill every field"
    cfg = eval(init_src, ns)
    assert "dist_abi_version() == 9" in block and 'dist_abi_sizeof(b"dist_config")' in block
    h = ctypes.c_void_p()
    pcfg = ctypes.cast(ctypes.pointer(cfg), ctypes.POINTER(lib.Config))    # the stub's own struct type, handed over as the bytes it is
    assert l.dist_create(pcfg, ctypes.byref(h)) == 0
    assert l.dist_param_count(h, 0) == 383
    l.dist_destroy(h)
    # garbage in the field an old binding would not have filled is refused, not interpreted
    for bad in (32, -1, 1 << 20, 16, 17):
        cfg.vit_fp8 = bad
        assert l.dist_create(pcfg, ctypes.byref(h)) == -1, bad
    cfg.vit_fp8 = 31
    cfg.dtype = 0                                      # fp8 spatial branch on an fp32 engine
    assert l.dist_create(pcfg, ctypes.byref(h)) == -1
    # DIST.SELECTED_LAYERS as a bit mask: a subset gives that many DiST layers in the parameter table; a block beyond the ViT is refused
    cfg.vit_fp8, cfg.dtype = 0, 1
    cfg.selected_mask = 0b101000000101                 # blocks 0, 2, 9, 11 of 12
    assert l.dist_create(pcfg, ctypes.byref(h)) == 0
    names = [l.dist_param_name(h, 0, i).decode() for i in range(l.dist_param_count(h, 0))]
    assert sum(n.startswith("dist_net.input_linears.") and n.endswith(".weight") for n in names) == 4
    assert "dist_net.integration_nets.3.ln.weight" in names and "dist_net.integration_nets.4.ln.weight" not in names
    l.dist_destroy(h)
    cfg.selected_mask = 1 << 12
    assert l.dist_create(pcfg, ctypes.byref(h)) == -1
    # MLP ratios as hidden widths: the parameter table follows them; widths that are no multiple of 8 are refused
    cfg.selected_mask, cfg.temporal_hidden, cfg.integration_hidden = 0, 192, 768
    assert l.dist_create(pcfg, ctypes.byref(h)) == 0
    dims = {l.dist_param_name(h, 0, i).decode(): tuple(l.dist_param_dim(h, 0, i, d) for d in range(l.dist_param_ndim(h, 0, i))) for i in range(l.dist_param_count(h, 0))}
    assert dims["dist_net.temporal_nets.0.temporal_net.c_fc1.weight"] == (192, 96, 3, 1, 1) and dims["dist_net.temporal_nets.0.temporal_net.c_fc2.weight"] == (96, 192, 1, 3, 3)
    assert dims["dist_net.integration_nets.0.ffn.c_fc.weight"] == (768, 384) and dims["dist_net.integration_nets.0.ffn.c_proj.weight"] == (384, 768)
    l.dist_destroy(h)
    cfg.temporal_hidden = 100
    assert l.dist_create(pcfg, ctypes.byref(h)) == -1


def test_engine_tables_without_gpu():
    """dist_create / parameter tables are host-only: names, shapes, offsets, groups match the reference layout."""
    from dist_amd import lib, synth
    from dist_amd.engine import config_from_geometry
    l = lib.load()
    g = synth.geometry("b16_8+16f")
    cfg = config_from_geometry(g, 32, torch.bfloat16)
    h = ctypes.c_void_p()
    assert l.dist_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    try:
        shapes = synth.dist_net_shapes(g)
        n = l.dist_param_count(h, 0)
        assert n == 383 and l.dist_param_total(h, 0) == 19001184
        seen, end = set(), 0
        for i in range(n):
            name = l.dist_param_name(h, 0, i).decode()
            shape = tuple(l.dist_param_dim(h, 0, i, d) for d in range(l.dist_param_ndim(h, 0, i)))
            assert shape == tuple(shapes[name]), name
            assert l.dist_param_offset(h, 0, i) == end
            end += int(np.prod(shape))
            seen.add(name)
        assert seen == set(shapes)
        # the two Linears of an IntegrationNetwork that read the same normalised rows keep their weights, and their biases, side by side: the fused
        # backward takes both weight gradients as ONE [Ci + C4][Ci] GEMM (engine.hip; the engine falls back to two GEMMs if this ever stops holding)
        off = {l.dist_param_name(h, 0, i).decode(): l.dist_param_offset(h, 0, i) for i in range(n)}
        for i in range(g.layers):
            p = f"dist_net.integration_nets.{i}."
            assert off[p + "temporal_ffn.c_fc1.weight"] == off[p + "ffn.c_fc.weight"] + g.Ci * g.Ci
            assert off[p + "temporal_ffn.c_fc1.bias"] == off[p + "ffn.c_fc.bias"] + g.Ci
        vs = synth.visual_shapes(g)
        assert {l.dist_param_name(h, 1, i).decode() for i in range(l.dist_param_count(h, 1))} == set(vs)
        assert l.dist_param_total(h, 1) == sum(int(np.prod(s)) for s in vs.values()) == 86192640
        assert l.dist_workspace_bytes(h) > 4 << 30 and l.dist_packed_bytes(h) > 0
        # nothing is bound: compute entry points must refuse, not crash
        assert l.dist_vit_forward(h, ctypes.c_void_p(8), 1, None) == lib.load().dist_vit_forward(h, ctypes.c_void_p(8), 1, None) != 0
    finally:
        l.dist_destroy(h)
    bad = config_from_geometry(g, 32, torch.bfloat16)
    bad.frames = 15                                   # not divisible by alpha
    assert l.dist_create(ctypes.byref(bad), ctypes.byref(h)) == -1


def test_engine_refuses_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dist_amd import lib, synth
    from dist_amd.engine import Engine, config_from_geometry
    with pytest.raises(lib.DistError):
        Engine(config_from_geometry(synth.geometry("tiny"), 2, torch.float32))


def test_synth_is_deterministic_and_well_scaled():
    from dist_amd import synth
    a = synth.uniform("x", (1000,), 0)
    b = synth.uniform("x", (1000,), 0)
    assert (a == b).all() and a.dtype == np.float32
    assert not (a == synth.uniform("x", (1000,), 1)).all() and not (a == synth.uniform("y", (1000,), 0)).all()
    # known-answer: pins the hash itself (values are regenerated on the GPU box)
    np.testing.assert_array_equal(synth.uniform("golden", (4,), 7), np.array(synth.uniform("golden", (8,), 7)[:4]))
    g = synth.geometry("tiny")
    t, y = synth.soft_target(g, 5)
    np.testing.assert_allclose(t.sum(1), 1.0, rtol=1e-5)
    tf = synth.text_features(g)
    np.testing.assert_allclose(np.linalg.norm(tf, axis=1), 1.0, rtol=1e-5)
    v = synth.video(g, 1)
    assert abs(v.std() - 1.0) < 0.02


WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from dist_amd.utils import distributed as du
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = port
du.init_process_group(rank, world, 0, backend="gloo")
assert du.get_world_size() == world and du.get_rank() == rank
assert du.is_master_proc() == (rank == 0)
# metric reduce: one packed collective for (loss, top1, top5) (reference runs/train.py:176-178)
loss, t1, t5 = du.all_reduce([torch.tensor(1.0 + rank), torch.tensor(10.0 * rank), torch.tensor(3.0)])
assert abs(float(loss) - (1.0 + (world - 1) / 2)) < 1e-6 and abs(float(t1) - 10.0 * (world - 1) / 2) < 1e-6 and float(t5) == 3.0
s, = du.all_reduce([torch.tensor([1.0, 2.0]) * (rank + 1)], average=False)
assert s.tolist() == [sum(range(1, world + 1)), 2 * sum(range(1, world + 1))]
# eval gather (reference runs/test.py:133-135): concatenated on dim 0 in rank order
p, l = du.all_gather([torch.full((2, 3), float(rank)), torch.tensor([rank, rank])])
assert p.shape == (2 * world, 3) and p[::2, 0].tolist() == [float(r) for r in range(world)] and l.tolist() == sum([[r, r] for r in range(world)], [])
# flat gradient reduce with the average folded into grad_scale (GradReducer semantics)
flat = torch.arange(8, dtype=torch.float32) * (rank + 1)
torch.distributed.all_reduce(flat)
assert torch.allclose(flat * (1.0 / world), torch.arange(8, dtype=torch.float32) * (world + 1) / 2)
m = du.all_reduce_max(torch.tensor([float(rank)]))
assert float(m) == world - 1
du.synchronize(); du.destroy()
print("OK", rank)
'''


def test_distributed_wrappers_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o


def test_distributed_single_process_defaults():
    from dist_amd.utils import distributed as du
    assert du.get_world_size() == 1 and du.get_rank() == 0 and du.is_master_proc()
    t = [torch.tensor(2.0)]
    assert du.all_reduce(t)[0] is t[0] and du.all_gather(t)[0] is t[0]


def test_grad_reducer_bucketing_logic():
    """slices arrive in descending order; adjacent ones are coalesced into buckets of >= bucket size."""
    from dist_amd.utils import distributed as du
    r = du.GradReducer.__new__(du.GradReducer)
    r.world, r.bucket_elems, r._pending, r.n_collectives = 2, 300, None, 0
    sent = []
    r._send = lambda b, e: sent.append((b, e))
    for sl in [(800, 1000), (700, 800), (600, 700), (500, 600), (400, 500), (100, 400), (0, 100)]:
        r._on_slice(*sl)
    if r._pending:
        r._send(*r._pending)
    assert sent == [(700, 1000), (400, 700), (100, 400), (0, 100)]
    assert sorted(sent)[0][0] == 0 and sum(e - b for b, e in sent) == 1000


def test_result_changing_knobs_exist_only_in_a_measure_build():
    """49 getenv knobs lived in the product library, several of which changed RESULTS (DIST_AMD_SKIP, DIST_AMD_DUMMY*,
    DIST_AMD_TN_SKIP_REDUCE, *_DBG).  Now: one place reads the environment (csrc/common.h), the result-changing names go through
    dist_measure_knob - a constant unless the library is built with -DDIST_AMD_MEASURE - and the shipped library says which kind it is."""
    import glob
    import re
    from dist_amd import lib
    l = lib.load()
    assert l.dist_measure_build() == 0                          # the in-tree library is the product build
    csrc = os.path.join(ROOT, "dist_amd", "csrc")
    srcs = glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "measure", "*.hip")) + glob.glob(os.path.join(csrc, "measure", "*.inl"))
    for f in srcs:
        assert "getenv" not in open(f).read() or f.endswith("common.h"), f     # only common.h's dist_knob touches the environment
    txt = "".join(open(f).read() for f in srcs if not f.endswith("common.h"))
    measure_only = {"DIST_AMD_SKIP", "DIST_AMD_DUMMY", "DIST_AMD_DUMMY_REPS", "DIST_AMD_TN_SKIP_REDUCE", "DIST_AMD_ATTN_DBG", "DIST_AMD_INTEG_DBG",
                    "DIST_AMD_TNET_DBG", "DIST_AMD_TNET_BWD_NOREDUCE", "DIST_AMD_PP_DBG"}
    sel = set(re.findall(r'dist_knob\("(DIST_AMD_[A-Z0-9_]+)"', txt))
    mea = set(re.findall(r'dist_measure_knob\("(DIST_AMD_[A-Z0-9_]+)"', txt))
    assert mea == measure_only and not (sel & measure_only), (mea ^ measure_only, sel & measure_only)
    # round 5: the product library reads at most 15 selectors - one per fused / fast kernel that has a fallback, the grid caps, the
    # serial-order switch; every A/B reference of a measured-and-rejected variant is a DIST_AB_KNOB (a constant outside the timing-only library)
    ab = set(re.findall(r'DIST_AB_KNOB\("(DIST_AMD_[A-Z0-9_]+)"', txt))
    assert len(sel) <= 15 and not (sel & ab) and len(ab) >= 20, (sorted(sel), sorted(sel & ab))
    assert sel <= lib.PRODUCT_ENV and not (lib.PRODUCT_ENV & (ab | mea))
    # round 6: the measure-only kernels are not sources of the product library, and a process that sets a measurement knob against the
    # product library is refused by the loader (it would time the same kernels under two labels)
    from dist_amd import build
    assert not any("measure" in x or "gemm_pp" in x for x in build.SOURCES) and all(x.startswith("measure") for x in build.MEASURE_SOURCES)
    r = subprocess.run([sys.executable, "-c", "from dist_amd import lib; lib.load()"], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, DIST_AMD_FAST_TILES="2"), timeout=300)
    assert r.returncode != 0 and "measurement knobs exist only in the timing-only library" in r.stderr, r.stderr[-400:]
    common = open(os.path.join(csrc, "common.h")).read()
    assert "#define DIST_AB_KNOB(name, dflt) (dflt)" in common
    assert "#ifdef DIST_AMD_MEASURE" in common and "inline int dist_measure_knob(const char*, int dflt) { return dflt; }" in common
