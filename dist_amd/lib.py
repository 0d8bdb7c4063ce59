"""ctypes binding of libdist_amd.so (include/dist_amd.h).  No CPU fallback: every
compute entry point raises if the HIP library is missing or a call fails."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIST_AMD_LIB") or os.path.join(_HERE, "csrc", "libdist_amd.so")   # DIST_AMD_LIB: another build of the same ABI (A/B measurements)

F32, BF16 = 0, 1
RM_PLAIN, RM_SHIFT, RM_SPATIAL, RM_STRIDED, RM_SKIPCLS = range(5)
OM_PLAIN, OM_DUP, OM_INSERTCLS, OM_SPLITCOLS, OM_HEADS = range(5)
QKV_ROWS, QKV_HEADS = 0, 1
EPI_BIAS, EPI_MULG, EPI_RES, EPI_ACT2, EPI_MULG_POST, EPI_LNFOLD, EPI_ROWSTATS, EPI_FP8 = 1, 2, 4, 8, 16, 32, 64, 128
EPI_FP8_ASCALAR, EPI_OUT8 = 256, 512


class RowMap(C.Structure):
    _fields_ = [("mode", C.c_int), ("p0", C.c_int), ("p1", C.c_int), ("sign", C.c_int)]


class OutMap(C.Structure):
    _fields_ = [("mode", C.c_int), ("p0", C.c_int), ("p1", C.c_int), ("p2", C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("C2", C.c_void_p),
                ("bias", C.c_void_p), ("res", C.c_void_p), ("aux", C.c_void_p),
                ("M", C.c_int64), ("N", C.c_int), ("K", C.c_int), ("taps", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int), ("ldc2", C.c_int), ("ldres", C.c_int), ("ldaux", C.c_int),
                ("amap", RowMap), ("omap", OutMap), ("flags", C.c_int), ("dtype", C.c_int), ("bias2", C.c_void_p), ("rowstats", C.c_void_p),
                ("a_scale", C.c_void_p), ("b_scale", C.c_void_p),
                ("C8", C.c_void_p), ("ldc8", C.c_int), ("out8_scale", C.c_void_p), ("out8_amax", C.c_void_p)]


class GemmTnArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int64), ("NI", C.c_int), ("K", C.c_int), ("taps", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("amap", RowMap), ("bmap", RowMap),
                ("so_i", C.c_int64), ("so_tap", C.c_int64), ("so_outer", C.c_int64), ("inner", C.c_int),
                ("dtype", C.c_int), ("use_tr", C.c_int), ("colsum", C.c_void_p), ("partial", C.c_void_p), ("partial_elems", C.c_int64),
                ("split_c", C.c_int), ("out2", C.c_void_p), ("so_i2", C.c_int64), ("colsum2", C.c_void_p), ("max_blocks", C.c_int)]


class LnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("y2", C.c_void_p),
                ("w", C.c_void_p), ("b", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("addend", C.c_void_p), ("addend_period", C.c_int),
                ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("rows", C.c_int64), ("C", C.c_int), ("dtype", C.c_int), ("eps", C.c_float)]


class LnBwdArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("dy", C.c_void_p), ("w", C.c_void_p), ("dy2", C.c_void_p), ("w2", C.c_void_p),
                ("dx", C.c_void_p), ("accumulate_dx", C.c_int),
                ("dw", C.c_void_p), ("db", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p),
                ("rows", C.c_int64), ("C", C.c_int), ("dtype", C.c_int), ("dx_add", C.c_void_p), ("dx_copy", C.c_void_p),
                ("partial", C.c_void_p), ("partial_elems", C.c_int64)]


class TnetArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("X", "W1", "W2", "b1", "b2", "ln_w", "ln_b", "z", "p", "Xp", "U", "V", "mean", "rstd")] + \
               [(n, C.c_int) for n in ("clips", "T", "G", "Ct", "tk", "dtype")] + [("eps", C.c_float)]


class TnetBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dp", "z", "X", "mean", "rstd", "ln_w", "W1b", "W2b", "dz", "dX", "dgamma", "dbeta", "scratch")] + \
               [("scratch_elems", C.c_int64)] + [(n, C.c_int) for n in ("clips", "T", "G", "Ct", "tk", "dtype", "phase")]


class IntegArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Mp", "W1", "W2", "W3", "b1", "b2", "b3", "ln_w", "ln_b", "ln_t_w", "ln_t_b", "R", "Na", "Nb", "mean", "rstd",
                                          "zf_h2", "hf_g2", "h1")] + \
               [(n, C.c_int) for n in ("clips", "t", "L", "Ci", "C4", "tk", "dtype")] + [("eps", C.c_float)] + \
               [(n, C.c_void_p) for n in ("t2i_M", "t2i_Xp", "t2i_W", "t2i_bias", "t2i_cls", "Mp_out", "i2t_W", "i2t_bias", "i2t_Xnext", "Xhat")]


class IntegUnfoldArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ffn_fc_w", "ln_w", "ln_b", "d_ffn_fc_w", "d_ffn_fc_b", "d_ln_w", "d_ln_b",
                                          "tf_fc1_w", "ln_t_w", "ln_t_b", "d_tf_fc1_w", "d_tf_fc1_b", "d_ln_t_w", "d_ln_t_b")] + [("Ci", C.c_int), ("C4", C.c_int)] + \
               [(n, C.c_void_p) for n in ("g_ffn_fc_w", "g_ffn_fc_b", "g_tf_fc1_w", "g_tf_fc1_b")]


class IntegPackArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ffn_fc_w", "ffn_fc_b", "ln_w", "ln_b", "tf_fc1_w", "tf_fc1_b", "ln_t_w", "ln_t_b", "tf_fc2_w", "tf_fc2_b",
                                          "ffn_proj_w", "ffn_proj_b", "tf_proj_w", "tf_proj_b", "W1", "W2", "W3", "b1", "b2", "b3")] + \
               [("Ci", C.c_int), ("C4", C.c_int)] + [(n, C.c_void_p) for n in ("B1", "B2", "B3", "t2i_w", "Wt", "i2t_w", "Wi", "W4", "W5")]


class IntegBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dR", "zf_h2", "Xhat", "rstd", "B1", "B2", "B3", "dzf_dh2", "dh1", "dMp", "dM_copy")] + \
               [(n, C.c_int) for n in ("add_dR", "clips", "t", "L", "Ci", "C4", "tk", "dtype")] + \
               [("ld_dzf", C.c_int), ("dh2", C.c_void_p), ("ld_dh2", C.c_int), ("ld_dh1", C.c_int)] + \
               [(n, C.c_void_p) for n in ("i2t_dXnext", "i2t_B", "i2t_dY", "t2i_B", "t2i_p", "t2i_dp", "t2i_dcls")] + [("dM_cls_only", C.c_int)]


class AdamwSeg(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("lr", C.c_float), ("weight_decay", C.c_float)]


class Config(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "dtype", "batch", "frames", "alpha", "resolution", "patch", "width", "layers",
        "integration_dim", "temporal_dim", "temporal_kernel", "temporal_patch", "int_temporal_div",
        "ada_layers", "num_classes", "embed_dim", "use_tr", "vit_fp8", "temporal_hidden", "integration_hidden", "selected_mask")]


GRAD_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)

ABI_VERSION = 9    # include/dist_amd.h DIST_ABI_VERSION: bumped on every struct-layout change
ABI_MIRRORS = (("dist_gemm_args", GemmArgs), ("dist_gemm_tn_args", GemmTnArgs), ("dist_ln_args", LnArgs), ("dist_ln_bwd_args", LnBwdArgs),
               ("dist_adamw_seg", AdamwSeg), ("dist_config", Config), ("dist_rowmap", RowMap), ("dist_outmap", OutMap), ("dist_tnet_args", TnetArgs), ("dist_tnet_bwd_args", TnetBwdArgs),
               ("dist_integ_args", IntegArgs), ("dist_integ_pack_args", IntegPackArgs), ("dist_integ_unfold_args", IntegUnfoldArgs), ("dist_integ_bwd_args", IntegBwdArgs))


class DistError(RuntimeError):
    pass


_lib = None


def _sig(lib, name, argtypes=None, restype=None):
    try:
        fn = getattr(lib, name)
    except AttributeError:
        raise DistError(f"libdist_amd.so does not export {name} (stale build? re-run build())")
    if argtypes is not None:
        fn.argtypes = argtypes
    if restype is not None or name.endswith("destroy"):
        fn.restype = restype


# every DIST_AMD_* variable the PRODUCT library or its Python host side reads (csrc/common.h dist_knob(); bench.py / utils/distributed.py / build.py)
PRODUCT_ENV = frozenset("DIST_AMD_" + k for k in (
    "ATTN_FULLROW CONV9 FAST_8P INTEG_BWD_FUSED INTEG_FUSED KEEP_MID LNFOLD NT_DMA ROWSTATS SERIAL TN8P TN8P_BLOCKS TNET_BWD_FUSED TNET_FUSED TN_BLOCKS "
    "LIB BACKEND FORCE_REDUCER VIT_SPLIT MAIN_PRIO HOST_INPUT BUILD_DEFS REDUCER_DTYPE REDUCER_MODE SKIP_SLOW_ORACLE ALLOW_INERT HOST_INPUT").split())


def load():
    """Load the HIP library or fail loudly (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DistError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  dist_amd has no CPU fallback.")
    # torch must be imported first: its bundled libamdhip64 has to be the HIP runtime of the process
    # (device pointers come from torch's allocator); loading ours first binds a second runtime copy.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    # a stale or foreign build (DIST_AMD_LIB) must not be driven through mirrors of another layout: host-only checks, no GPU needed
    _sig(lib, "dist_abi_version", argtypes=[])
    _sig(lib, "dist_abi_sizeof", argtypes=[C.c_char_p])
    if lib.dist_abi_version() != ABI_VERSION:
        raise DistError(f"{LIB_PATH}: ABI version {lib.dist_abi_version()}, this binding is written for {ABI_VERSION} (stale build? re-run build())")
    for cname, mirror in ABI_MIRRORS:
        if lib.dist_abi_sizeof(cname.encode()) != C.sizeof(mirror):
            raise DistError(f"{LIB_PATH}: sizeof({cname}) = {lib.dist_abi_sizeof(cname.encode())}, the ctypes mirror has {C.sizeof(mirror)}")
    _sig(lib, "dist_measure_build", argtypes=[])
    # a measurement knob (DIST_AB_KNOB / dist_measure_knob in csrc/common.h) is a constant in the product library: a run that sets one against
    # libdist_amd.so would time the SAME kernels under different labels - refuse instead of producing a vacuous A/B
    if not lib.dist_measure_build() and os.environ.get("DIST_AMD_ALLOW_INERT") != "1":      # (ALLOW_INERT: the test that proves the C library ignores them)
        inert = sorted(k for k in os.environ if k.startswith("DIST_AMD_") and k not in PRODUCT_ENV)
        if inert:
            raise DistError(f"{', '.join(inert)}: measurement knobs exist only in the timing-only library (`. tools/measure_build.sh`, DIST_AMD_LIB); "
                            f"{LIB_PATH} is the product build and would ignore them")
    _sig(lib, "dist_strerror", restype=C.c_char_p)
    _sig(lib, "dist_strerror", argtypes=[C.c_int])
    _sig(lib, "dist_abi_sizeof", argtypes=[C.c_char_p])
    _sig(lib, "dist_last_error", restype=C.c_char_p)
    _sig(lib, "dist_last_error", argtypes=[C.c_void_p])
    _sig(lib, "dist_param_name", restype=C.c_char_p)
    _sig(lib, "dist_param_name", argtypes=[C.c_void_p, C.c_int, C.c_int])
    for fn in ("dist_param_dim", "dist_param_offset", "dist_param_total"):
        _sig(lib, fn, restype=C.c_int64)
    _sig(lib, "dist_param_dim", argtypes=[C.c_void_p, C.c_int, C.c_int, C.c_int])
    _sig(lib, "dist_param_offset", argtypes=[C.c_void_p, C.c_int, C.c_int])
    _sig(lib, "dist_param_total", argtypes=[C.c_void_p, C.c_int])
    _sig(lib, "dist_param_count", argtypes=[C.c_void_p, C.c_int])
    _sig(lib, "dist_param_ndim", argtypes=[C.c_void_p, C.c_int, C.c_int])
    _sig(lib, "dist_param_group", argtypes=[C.c_void_p, C.c_int])
    _sig(lib, "dist_workspace_bytes", restype=C.c_size_t)
    _sig(lib, "dist_workspace_bytes", argtypes=[C.c_void_p])
    _sig(lib, "dist_packed_bytes", restype=C.c_size_t)
    _sig(lib, "dist_packed_bytes", argtypes=[C.c_void_p])
    _sig(lib, "dist_create", argtypes=[C.POINTER(Config), C.POINTER(C.c_void_p)])
    _sig(lib, "dist_destroy", argtypes=[C.c_void_p])
    _sig(lib, "dist_destroy", restype=None)
    _sig(lib, "dist_bind", argtypes=[C.c_void_p] * 8)
    _sig(lib, "dist_pack_weights", argtypes=[C.c_void_p, C.c_int, C.c_void_p])
    _sig(lib, "dist_vit_forward", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_void_p])
    _sig(lib, "dist_vit_prefetch", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_vit_prefetch_layers", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_vit_adopt", argtypes=[C.c_void_p])
    _sig(lib, "dist_features_import", argtypes=[C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int, C.c_void_p])
    _sig(lib, "dist_branch_forward", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_branch_backward", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p])
    _sig(lib, "dist_loss", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_set_inference", argtypes=[C.c_void_p, C.c_int])
    _sig(lib, "dist_set_grad_ready_hook", argtypes=[C.c_void_p, GRAD_HOOK, C.c_void_p])
    _sig(lib, "dist_marks_enable", argtypes=[C.c_void_p, C.c_int])
    _sig(lib, "dist_marks_read", argtypes=[C.c_void_p, C.POINTER(C.c_float), C.c_int])
    _sig(lib, "dist_profile_begin", argtypes=[C.c_void_p])
    _sig(lib, "dist_profile_end", argtypes=[C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)])
    _sig(lib, "dist_debug_tensor", argtypes=[C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)])
    _sig(lib, "dist_op_gemm_nt", argtypes=[C.POINTER(GemmArgs), C.c_void_p])
    _sig(lib, "dist_op_temporal_net_fwd", argtypes=[C.POINTER(TnetArgs), C.c_void_p])
    _sig(lib, "dist_op_temporal_net_bwd", argtypes=[C.POINTER(TnetBwdArgs), C.c_void_p])
    _sig(lib, "dist_op_temporal_net_bwd_reduce", argtypes=[C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_integration_fwd", argtypes=[C.POINTER(IntegArgs), C.c_void_p])
    _sig(lib, "dist_op_integration_bwd", argtypes=[C.POINTER(IntegBwdArgs), C.c_void_p])
    _sig(lib, "dist_op_integration_unfold", argtypes=[C.POINTER(IntegUnfoldArgs), C.c_void_p])
    _sig(lib, "dist_op_integration_pack", argtypes=[C.POINTER(IntegPackArgs), C.c_void_p])
    _sig(lib, "dist_op_integration_pack_elems", argtypes=[C.c_int, C.c_int, C.c_int], restype=C.c_int64)
    _sig(lib, "dist_op_temporal_net_bwd_scratch", argtypes=[C.c_int, C.c_int, C.c_int], restype=C.c_int64)
    _sig(lib, "dist_op_ln_fold", argtypes=[C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p])
    _sig(lib, "dist_op_gemm_tn", argtypes=[C.POINTER(GemmTnArgs), C.c_void_p])
    _sig(lib, "dist_op_layernorm", argtypes=[C.POINTER(LnArgs), C.c_void_p])
    _sig(lib, "dist_op_ln_stats_from_partials", argtypes=[C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_layernorm_bwd", argtypes=[C.POINTER(LnBwdArgs), C.c_void_p])
    _sig(lib, "dist_op_layernorm_bwd_scratch", argtypes=[C.c_int64, C.c_int], restype=C.c_int64)
    _sig(lib, "dist_op_attention", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p])
    _sig(lib, "dist_op_attention_out8", argtypes=[C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p])
    _sig(lib, "dist_op_attention_fp8", argtypes=[C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_void_p])
    _sig(lib, "dist_op_xattn1q", argtypes=[C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p])
    _sig(lib, "dist_op_xattn1q_bwd", argtypes=[C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p])
    _sig(lib, "dist_op_patchify", argtypes=[C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p])
    _sig(lib, "dist_op_add", argtypes=[C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p])
    _sig(lib, "dist_op_gelu_bwd", argtypes=[C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p])
    _sig(lib, "dist_op_colsum", argtypes=[C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, RowMap, C.c_int, C.c_void_p])
    _sig(lib, "dist_op_logits_loss", argtypes=[C.c_void_p] * 10 + [C.c_int] * 4 + [C.c_void_p])
    _sig(lib, "dist_op_mixup", argtypes=[C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_void_p])
    _sig(lib, "dist_op_cutmix", argtypes=[C.c_void_p] + [C.c_int] * 8 + [C.c_void_p])
    _sig(lib, "dist_op_patchify_mixed", argtypes=[C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_float, C.c_float] + [C.c_int] * 4 + [C.c_void_p])
    _sig(lib, "dist_vit_mix_next", argtypes=[C.c_void_p, C.c_int, C.c_float, C.c_float] + [C.c_int] * 4)
    _sig(lib, "dist_op_mixup_target", argtypes=[C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_quant_rows_fp8", argtypes=[C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_fp8_scale_update", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p])
    _sig(lib, "dist_op_amax", argtypes=[C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_fp8_rowsum", argtypes=[C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_softmax_rows", argtypes=[C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_topk_correct", argtypes=[C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_ensemble_update", argtypes=[C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
    _sig(lib, "dist_op_adamw", argtypes=[C.c_void_p] * 5 + [C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p])
    _lib = lib
    return lib


def check(code, handle=None):
    if code != 0:
        lib = load()
        msg = lib.dist_strerror(code).decode()
        if handle is not None:
            extra = lib.dist_last_error(handle)
            if extra:
                msg += ": " + extra.decode()
        raise DistError(f"dist_amd error {code}: {msg}")


def exported_symbols():
    """Every `dist_*` function declared in include/dist_amd.h."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "dist_amd.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dist_[a-z0-9_]+)\s*\(", txt)))
