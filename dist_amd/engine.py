"""Python owner of one `dist_handle` (include/dist_amd.h, engine level).

torch is the allocator / stream provider only: this class allocates the flat fp32
parameter, gradient and AdamW moment buffers, the packed working-weight buffer and the
activation workspace as torch tensors, binds their raw pointers into the C engine and
then drives `dist_vit_forward / dist_branch_forward / dist_loss / dist_branch_backward /
dist_op_adamw` on torch's current HIP stream.  There is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from . import ops


def config_from_geometry(g, batch, dtype, use_tr=True, vit_fp8=0):
    """g: dist_amd.synth.Geometry (or any object with the same attributes).  vit_fp8: bit mask of the frozen-ViT GEMMs on e4m3 operands
    (1 in_proj, 2 out_proj, 4 c_fc, 8 c_proj; BASELINE config 5), bf16 engines only."""
    c = L.Config()
    c.dtype = L.BF16 if dtype == torch.bfloat16 else L.F32
    c.batch, c.frames, c.alpha = batch, g.T, g.alpha
    c.resolution, c.patch, c.width, c.layers = g.res, g.patch, g.d, g.layers
    c.integration_dim, c.temporal_dim = g.Ci, g.Ct
    c.temporal_kernel, c.temporal_patch = g.tk, g.tpatch
    c.int_temporal_div = int(round(1.0 / g.int_t_ratio))
    c.ada_layers, c.num_classes, c.embed_dim = g.ada, g.K, g.E
    c.use_tr = int(use_tr)
    c.vit_fp8 = int(vit_fp8)
    c.temporal_hidden = 0 if getattr(g, "Ch", g.Ct) == g.Ct else int(g.Ch)          # DIST.TEMPORAL_CONV_MLP_RATIO / INTEGRATION_MLP_RATIO: 0 = ratio 1
    c.integration_hidden = 0 if getattr(g, "Cf", g.Ci) == g.Ci else int(g.Cf)
    sel = tuple(getattr(g, "selected", None) or range(g.layers))           # DIST.SELECTED_LAYERS -> bit mask (0 = every block)
    c.selected_mask = 0 if sel == tuple(range(g.layers)) else sum(1 << i for i in sel)
    return c


class Engine:
    def __init__(self, cfg, device="cuda"):
        if not torch.cuda.is_available():
            raise L.DistError("dist_amd.Engine needs a GPU: the HIP library is the only implementation")
        self.lib = L.load()
        self.cfg = cfg
        # DIST.SELECTED_LAYERS: the ViT blocks whose outputs feed the branch (DiST layer k reads block selected[k])
        self.selected = [i for i in range(cfg.layers) if (not cfg.selected_mask) or (cfg.selected_mask >> i) & 1]
        self.device = torch.device(device)
        self.dtype = torch.bfloat16 if cfg.dtype == L.BF16 else torch.float32
        h = C.c_void_p()
        L.check(self.lib.dist_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.tables = [self._table(0), self._table(1)]
        n0 = self.lib.dist_param_total(h, 0)
        n1 = self.lib.dist_param_total(h, 1)
        dev = self.device
        self.theta = torch.zeros(n0, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n0, dtype=torch.float32, device=dev)
        self.visual = torch.zeros(n1, dtype=torch.float32, device=dev)
        self.logit_scale = torch.full((1,), float(np.log(1 / 0.07)), dtype=torch.float32, device=dev)
        self.dlogit_scale = torch.zeros(1, dtype=torch.float32, device=dev)
        self.packed = torch.empty(self.lib.dist_packed_bytes(h), dtype=torch.uint8, device=dev)
        self.workspace = torch.empty(self.lib.dist_workspace_bytes(h), dtype=torch.uint8, device=dev)
        L.check(self.lib.dist_bind(h, self.theta.data_ptr(), self.grads.data_ptr(), self.visual.data_ptr(),
                                   self.logit_scale.data_ptr(), self.dlogit_scale.data_ptr(),
                                   self.packed.data_ptr(), self.workspace.data_ptr()), h)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.step_count = 0
        self._segs = None
        self._segs_key = None
        self._text = None
        self._pf_video = None
        self._feat_stamp = 0                 # bumped whenever the current feature slot receives another pass
        self._branch_stamp = 0               # bumped by every branch forward: the loss path checks that the logits it is given are the last ones
        self._cur_video = None
        self._cur_ver = self._pf_ver = -1
        self.b = 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dist_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- parameter tables -------------------------------------------------------------------
    def _table(self, kind):
        lib, h = self.lib, self.h
        out = {}
        for i in range(lib.dist_param_count(h, kind)):
            name = lib.dist_param_name(h, kind, i).decode()
            shape = tuple(lib.dist_param_dim(h, kind, i, d) for d in range(lib.dist_param_ndim(h, kind, i)))
            out[name] = (lib.dist_param_offset(h, kind, i), shape, lib.dist_param_group(h, i) if kind == 0 else -1)
        return out

    def _flat(self, kind):
        return self.theta if kind == 0 else self.visual

    def view(self, name, grad=False):
        for kind in (0, 1):
            if name in self.tables[kind]:
                off, shape, _ = self.tables[kind][name]
                n = int(np.prod(shape)) if shape else 1
                base = (self.grads if grad else self.theta) if kind == 0 else self.visual
                return base[off:off + n].view(shape)
        if name == "logit_scale":
            return (self.dlogit_scale if grad else self.logit_scale).view(())
        raise KeyError(name)

    def load_state_dict(self, sd, strict=True):
        """sd: reference state-dict names -> tensor / ndarray (visual.*, dist_net.*, logit_scale).
        Text-tower keys are ignored (the text features are an input)."""
        missing = []
        for kind in (0, 1):
            for name, (off, shape, _) in self.tables[kind].items():
                if name not in sd:
                    missing.append(name)
                    continue
                t = torch.as_tensor(sd[name]).to(torch.float32).reshape(-1)
                n = int(np.prod(shape)) if shape else 1
                if t.numel() != n:
                    raise L.DistError(f"{name}: expected {shape}, got {tuple(torch.as_tensor(sd[name]).shape)}")
                self._flat(kind)[off:off + n].copy_(t)
        if "logit_scale" in sd:
            self.logit_scale.fill_(float(torch.as_tensor(sd["logit_scale"])))
        if strict and missing:
            raise L.DistError(f"missing keys: {missing[:8]}{'...' if len(missing) > 8 else ''}")
        self.pack(3)
        return missing

    def state_dict(self):
        sd = {name: self.view(name).detach().clone() for kind in (0, 1) for name in self.tables[kind]}
        sd["logit_scale"] = self.logit_scale.view(()).clone()
        return sd

    def pack(self, what=3):
        L.check(self.lib.dist_pack_weights(self.h, what, ops._stream()), self.h)
        self._packed_version = (self.theta._version, self.visual._version)

    def sync_packed(self, param_version=0):
        """Re-pack the working weight copies if the master weights were written since the last pack
        (`param_version`: any monotone stamp of the nn.Parameter views, e.g. the sum of their `_version`)."""
        ver = (self.theta._version, self.visual._version, param_version)
        last = getattr(self, "_sync_version", None)
        if last is None or last != ver:
            self.pack(3)
            self._sync_version = (self.theta._version, self.visual._version, param_version)

    # ---- the hot path ----------------------------------------------------------------------
    def vit_mix_next(self, kind="none", lam=1.0, box=None):
        """The next ViT pass that starts gathers its patch rows from the batch AS IF mixed (dist_vit_mix_next): kind "mixup" | "cutmix" | "none",
        `lam` / `box` as dist_amd.dataset.utils.mixup.MixPlan carries them.  The frames themselves are not written."""
        yl, yh, xl, xh = box if box is not None else (0, 0, 0, 0)
        L.check(self.lib.dist_vit_mix_next(self.h, ops.MIX_KINDS[kind], ops._f32(lam), ops._f32(1.0 - lam), int(yl), int(yh), int(xl), int(xh)), self.h)

    def _deferred_mix(self, video):
        """A MixPlan that Mixup(fuse=True) left on the clip tensor instead of mixing it (TRAIN.FUSE_MIXUP): handed to the pass that is about to read the clip."""
        plan = getattr(video, "_dist_mix", None)
        if plan is not None:
            self.vit_mix_next(plan.kind, plan.lam, plan.box)        # (cutmix: lam only builds the soft target, the box is what moves)
            del video._dist_mix

    def vit_forward(self, video):
        assert video.dtype == torch.float32 and video.is_contiguous() and video.is_cuda
        self._deferred_mix(video)
        self.b = video.shape[0]
        L.check(self.lib.dist_vit_forward(self.h, video.data_ptr(), self.b, ops._stream()), self.h)
        self._cur_video, self._cur_ver = video, video._version
        self._feat_stamp = getattr(self, "_feat_stamp", 0) + 1      # (which pass the current feature slot holds: CLIP's lazy img_logits checks it)

    def import_features(self, mid_feat, video):
        """The caller's own frozen-ViT features instead of a ViT pass (dist_features_import): `mid_feat` = one tensor per ViT block in the
        reference's layout [L, b*t, width] (fp32 or bf16, CUDA, contiguous), `video` [b,3,T,H,W] fp32.  What the reference's
        DiSTNetwork.forward reads from input['mid_feat']['img'] / input['images'] (dist.py:222-247)."""
        assert video.dtype == torch.float32 and video.is_contiguous() and video.is_cuda
        nl = self.cfg.layers
        sel = self.selected
        if isinstance(mid_feat, dict):                      # {ViT block: tensor} as the reference's input['mid_feat']['img']
            missing = [i for i in sel if i not in mid_feat]
            if missing:
                raise L.DistError(f"import_features: no tensor for the selected blocks {missing}")
            mid_feat = [mid_feat[i] for i in sel]
        if len(mid_feat) != len(sel):
            raise L.DistError(f"import_features needs one tensor per selected block ({len(sel)}), got {len(mid_feat)}")
        dt = mid_feat[0].dtype
        if dt not in (torch.float32, torch.bfloat16):
            raise L.DistError(f"mid_feat dtype {dt}: float32 or bfloat16")
        b = video.shape[0]
        t_, Lt = self.cfg.frames // self.cfg.alpha, (self.cfg.resolution // self.cfg.patch) ** 2 + 1
        keep = []
        for i, m in enumerate(mid_feat):
            if tuple(m.shape) != (Lt, b * t_, self.cfg.width) or m.dtype != dt or not m.is_cuda:
                raise L.DistError(f"mid_feat[{i}]: expected a CUDA {dt} tensor of shape {(Lt, b * t_, self.cfg.width)}, got {m.dtype} {tuple(m.shape)}")
            keep.append(m.contiguous())
        ptrs = (C.c_void_p * nl)()                           # indexed by ViT block; NULL for blocks that are not selected
        for blk, m in zip(sel, keep):
            ptrs[blk] = m.data_ptr()
        self.b = b
        L.check(self.lib.dist_features_import(self.h, ptrs, L.F32 if dt == torch.float32 else L.BF16, video.data_ptr(), b, ops._stream()), self.h)
        self._cur_video, self._cur_ver = None, -1          # the slot no longer holds the pass of a known clip tensor
        self._feat_stamp = getattr(self, "_feat_stamp", 0) + 1

    def has_features_for(self, video):
        """True when the current feature slot already holds the frozen-ViT pass of exactly this tensor (same object,
        not written since): it was prefetched during the previous step and adopted."""
        return self._cur_video is video and video._version == self._cur_ver

    def vit_prefetch(self, video, layer_end=None, stream=None):
        """Frozen ViT of the NEXT batch into the spare feature slot, on the handle's lowest-priority prefetch stream, behind
        everything already queued on the current stream (dist_vit_prefetch).  `video` must stay alive until `vit_adopt`.
        layer_end < layers issues only the first layers; `vit_prefetch_more` continues the pass later in the step."""
        assert video.dtype == torch.float32 and video.is_contiguous() and video.is_cuda
        self._deferred_mix(video)
        self._pf_video, self._pf_ver = video, video._version
        le = self.cfg.layers if layer_end is None else int(layer_end)
        # `stream`: a torch stream of the caller's that carries the pass instead of the handle's own prefetch stream (the C ABI's `stream` argument) - e.g. the
        # one that also carries the host -> device copies of the batches (dist_amd/utils/staging.py): the number of ACTIVE streams stays at four
        self._pf_stream = C.c_void_p(stream.cuda_stream) if stream is not None else None
        L.check(self.lib.dist_vit_prefetch_layers(self.h, video.data_ptr(), video.shape[0], le, self._pf_stream, ops._stream()), self.h)

    def vit_prefetch_more(self, layer_end=None):
        le = self.cfg.layers if layer_end is None else int(layer_end)
        L.check(self.lib.dist_vit_prefetch_layers(self.h, None, 0, le, getattr(self, "_pf_stream", None), ops._stream()), self.h)

    def vit_adopt(self):
        """The prefetched batch becomes the current one (as after `vit_forward` of it)."""
        L.check(self.lib.dist_vit_adopt(self.h), self.h)
        self.b = self._pf_video.shape[0]
        self._cur_video, self._cur_ver, self._pf_video = self._pf_video, self._pf_ver, None
        self._feat_stamp = getattr(self, "_feat_stamp", 0) + 1

    def set_inference(self, on=True):
        """forward passes keep nothing for a backward pass (dist_set_inference); `backward` then raises until a training-mode forward ran"""
        L.check(self.lib.dist_set_inference(self.h, int(bool(on))), self.h)

    def branch_forward(self, text_features):
        assert text_features.dtype == torch.float32 and text_features.is_contiguous()
        self._text = text_features            # keep alive: the engine borrows the pointer until backward
        self._branch_stamp = getattr(self, "_branch_stamp", 0) + 1     # which forward the engine's video embedding belongs to (losses.py checks it)
        logits = torch.empty(self.b, self.cfg.num_classes, dtype=torch.float32, device=self.device)
        vid = torch.empty(self.b, self.cfg.embed_dim, dtype=torch.float32, device=self.device)
        L.check(self.lib.dist_branch_forward(self.h, text_features.data_ptr(), self.b, logits.data_ptr(), vid.data_ptr(), ops._stream()), self.h)
        return logits, vid

    def loss(self, soft_target):
        assert soft_target.dtype == torch.float32 and soft_target.is_contiguous()
        loss = torch.empty((), dtype=torch.float32, device=self.device)
        dlogits = torch.empty(self.b, self.cfg.num_classes, dtype=torch.float32, device=self.device)
        L.check(self.lib.dist_loss(self.h, soft_target.data_ptr(), self.b, loss.data_ptr(), dlogits.data_ptr(), ops._stream()), self.h)
        return loss, dlogits

    def backward(self, dlogits, zero_grads=True):
        assert dlogits.dtype == torch.float32 and dlogits.is_contiguous()
        L.check(self.lib.dist_branch_backward(self.h, dlogits.data_ptr(), self.b, int(zero_grads), ops._stream()), self.h)

    def forward_backward(self, video, text_features, soft_target):
        """One reference train step up to the gradients (runs/train.py:101-110)."""
        self.vit_forward(video)
        logits, _ = self.branch_forward(text_features)
        loss, dlogits = self.loss(soft_target)
        self.backward(dlogits)
        return loss, logits

    # ---- optimizer (models/utils/optimizer.py:67-73,138-186 as intended) ----------------------
    def adamw_step(self, lr, weight_decay, lr_mult=1.0, grad_scale=1.0):
        """lr: base lr of this iteration; groups 1 and 3 get `weight_decay`, groups 0, 2, 4 none;
        every DiST group uses lr * lr_mult (NEW_NET_LRMULT)."""
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.theta)
            self.exp_avg_sq = torch.zeros_like(self.theta)
        key = (float(lr), float(weight_decay), float(lr_mult))
        if self._segs_key != key:
            ent = []
            for name, (off, shape, grp) in self.tables[0].items():
                n = int(np.prod(shape)) if shape else 1
                ent.append((off, off + n, lr * lr_mult, weight_decay if grp in (1, 3) else 0.0))
            ent.sort()
            merged = []
            for e in ent:
                if merged and merged[-1][1] == e[0] and merged[-1][2:] == e[2:]:
                    merged[-1] = (merged[-1][0], e[1]) + e[2:]
                else:
                    merged.append(e)
            self._segs = ops.make_segs(merged, self.device)
            self._segs_key = key
        self.step_count += 1
        ops.adamw(self.theta, self.grads, self.exp_avg, self.exp_avg_sq, self._segs, self.step_count, grad_scale=grad_scale)
        self.pack(2)

    def set_grad_ready_hook(self, fn):
        """fn(begin, end) is called while backward is being enqueued, once per finished gradient slice."""
        if fn is None:
            self._hook = L.GRAD_HOOK(0)
        else:
            self._hook = L.GRAD_HOOK(lambda _user, b, e: fn(int(b), int(e)))
        L.check(self.lib.dist_set_grad_ready_hook(self.h, self._hook, None), self.h)

    MARKS = ("vit_begin", "vit_end", "fwd_begin", "fwd_mid", "fwd_end", "bwd_begin", "bwd_end", "step_end")

    def marks_enable(self, on=True):
        L.check(self.lib.dist_marks_enable(self.h, int(on)), self.h)

    def marks_read(self):
        """{mark: ms since this step's vit_begin} of the LAST step (device-side HIP events; synchronises)."""
        ms = (C.c_float * len(self.MARKS))()
        L.check(self.lib.dist_marks_read(self.h, ms, len(self.MARKS)), self.h)
        return dict(zip(self.MARKS, [float(v) for v in ms]))

    def profile_begin(self):
        L.check(self.lib.dist_profile_begin(self.h), self.h)

    def profile_end(self):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int()
        L.check(self.lib.dist_profile_end(self.h, C.byref(ms), C.byref(fl), C.byref(n)), self.h)
        return ms.value, fl.value, n.value

    def debug(self, name):
        ptr, rows, cols = C.c_void_p(), C.c_int64(), C.c_int()
        L.check(self.lib.dist_debug_tensor(self.h, name.encode(), C.byref(ptr), C.byref(rows), C.byref(cols)), self.h)
        n = rows.value * cols.value
        es = 2 if self.dtype == torch.bfloat16 else 4
        off = ptr.value - self.workspace.data_ptr()
        return self.workspace[off:off + n * es].view(self.dtype).view(rows.value, cols.value)
