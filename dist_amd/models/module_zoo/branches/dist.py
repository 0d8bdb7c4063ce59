"""DiST branch behind the reference's `DiSTNetwork` surface (reference models/module_zoo/branches/dist.py:165-247).

The module owns the trainable `dist_net.*` parameters (views into the engine's flat fp32 buffer, reference
names and shapes) and maps the config keys the reference reads (dist.py:19-38,51-61,71-76,170-190) onto
`dist_config`.  TemporalNet / IntegrationNetwork / Temporal2Integration / Integration2Temporal /
SpatialTemporalAdaPooling are not separate Python modules here: they are the kernel sequence of
dist_branch_forward / dist_branch_backward (dist_amd/csrc/engine.hip).
"""
import torch

from .... import lib as L
from ...base.clip import DiSTParams


def engine_config(cfg, width, layers, patch, resolution, embed_dim, batch, dtype):
    d = cfg.VIDEO.BACKBONE.DIST
    sel = [int(i) for i in d.SELECTED_LAYERS]
    # any subset of the ViT's blocks in increasing order (reference dist.py:170-190, 226: one DiST layer per selected block; every released yaml selects all)
    if not sel or sel != sorted(set(sel)) or sel[0] < 0 or sel[-1] >= layers or layers > 32:
        raise L.DistError(f"SELECTED_LAYERS must be an increasing list of blocks in [0, {layers}), got {sel}")
    if int(d.S_PATCH_SIZE) != int(patch):
        raise L.DistError(f"DIST.S_PATCH_SIZE {d.S_PATCH_SIZE} must equal the ViT patch {patch} "
                          "(the L/14 yamls of the reference carry 16, which fails there as well: SURVEY.md §0)")
    # hidden widths as the reference computes them (dist.py:20-25, 51-58: int(dim * ratio)); ratio 1 (every released yaml) runs the fused kernels
    ch, cf = int(int(d.TEMPORAL_DIM) * float(d.TEMPORAL_CONV_MLP_RATIO)), int(int(d.INTEGRATION_DIM) * float(d.INTEGRATION_MLP_RATIO))
    if ch <= 0 or ch % 8 or cf <= 0 or cf % 8:
        raise L.DistError(f"TEMPORAL_CONV_MLP_RATIO / INTEGRATION_MLP_RATIO give hidden widths {ch} / {cf}: multiples of 8 are needed")
    c = L.Config()
    c.dtype = L.BF16 if dtype == torch.bfloat16 else L.F32
    c.batch, c.frames, c.alpha = batch, int(cfg.DATA.NUM_INPUT_FRAMES), int(cfg.DATA.SPARSE_SAMPLE_ALPHA)
    c.resolution, c.patch, c.width, c.layers = resolution, patch, width, layers
    c.integration_dim, c.temporal_dim = int(d.INTEGRATION_DIM), int(d.TEMPORAL_DIM)
    c.temporal_kernel, c.temporal_patch = int(d.TEMPORAL_KERNEL_SIZE), int(d.T_PATCH_SIZE)
    c.int_temporal_div = int(round(1.0 / float(d.INTEGRATION_TEMPORAL_MLP_RATIO)))
    c.ada_layers, c.num_classes, c.embed_dim = int(d.ADA_POOLING_LAYERS), int(cfg.VIDEO.HEAD.NUM_CLASSES), embed_dim
    c.use_tr = 1
    # BASELINE config 5 (fp8 frozen spatial branch): VIDEO.BACKBONE.FP8_SPATIAL = bit mask of the frozen-ViT GEMMs on e4m3 operands
    # (15 = all four; absent / 0 = bf16).  Not a key of the reference's yamls: pass it as a trailing KEY VAL override.
    c.vit_fp8 = int(getattr(cfg.VIDEO.BACKBONE, "FP8_SPATIAL", 0) or 0) if dtype == torch.bfloat16 else 0
    c.temporal_hidden = 0 if ch == int(d.TEMPORAL_DIM) else ch
    c.integration_hidden = 0 if cf == int(d.INTEGRATION_DIM) else cf
    c.selected_mask = 0 if sel == list(range(layers)) else sum(1 << i for i in sel)
    return c


class DiSTNetwork(DiSTParams):
    def __init__(self, cfg, d_model, width, output_dim, engine=None):
        super().__init__()
        if engine is None:
            raise L.DistError("DiSTNetwork is constructed by CLIP with its engine (the HIP library is the only implementation)")
        self.cfg = cfg
        self.selected_layers = cfg.VIDEO.BACKBONE.DIST.SELECTED_LAYERS
        self.alpha = int(cfg.DATA.SPARSE_SAMPLE_ALPHA)
        self.num_frames = cfg.DATA.NUM_INPUT_FRAMES
        self._engine = [engine]          # not a submodule / buffer
        self._adopt(engine, "dist_net.", 0, requires_grad=True)
        self._init_like_reference()

    def _init_like_reference(self):
        """trunc_normal(0.02) weights, zero biases, unit LayerNorms (reference dist.py:204-220, 76-78, 120-122, 193-201)."""
        eng = self._engine[0]
        with torch.no_grad():
            for name, (off, shape, grp) in eng.tables[0].items():
                v = eng.view(name)
                if name.endswith(("ln.weight", "ln_1.weight", "ln_post.weight", "ln_temporal.weight",
                                  "ln_out_temp_cls_token.weight", "ln_out_spat_cls_token.weight")):
                    v.fill_(1.0)
                elif len(shape) == 1:
                    v.zero_()
                elif name == "dist_net.proj":
                    v.normal_(std=shape[0] ** -0.5)
                else:
                    torch.nn.init.trunc_normal_(v, std=0.02)
        eng.pack(2)

    def forward(self, input):
        """reference signature `forward(input: dict) -> (cls_x [b,E], input)` (dist.py:222-247).  The reference reads
        input['mid_feat']['img'][layer_id] ([L, b*t, width], one per selected layer) and input['images'] ([b*T,3,H,W] or [b,3,T,H,W]): when the
        caller supplies them they are copied into the engine's feature slot (dist_features_import) and the branch runs on THEM; without
        them the branch runs on the features of the engine's own frozen-ViT pass of the same batch (CLIP.forward runs dist_vit_forward
        first).  input['text_features'] as in the reference (clip.py:507).
        Two documented differences: the returned embedding is already L2-normalised (the reference normalises cls_x in the next statement of its
        caller, clip.py:511; the engine's head kernel does it in the same launch that forms the logits, which are left in input['logits_per_image']),
        and this direct call does not record an autograd graph - training goes through CLIP.forward (`_DistFunction`), whose backward is
        dist_branch_backward."""
        eng = self._engine[0]
        mid = input.get("mid_feat", {}).get("img") if isinstance(input.get("mid_feat"), dict) else None
        if mid is not None and all(i in mid for i in self.selected_layers):
            if "images" not in input:
                raise L.DistError("DiSTNetwork.forward: input['mid_feat'] needs input['images'] beside it (the temporal stem reads the frames, dist.py:225)")
            img = input["images"]
            if img.dim() == 4:                               # [b*T,3,H,W] as the reference backbone passes it
                bt, c, h, w = img.shape
                img = img.view(bt // self.num_frames, self.num_frames, c, h, w).permute(0, 2, 1, 3, 4)
            eng.import_features({int(i): mid[i] for i in self.selected_layers}, img.contiguous().float())
        tf = input["text_features"].float().contiguous()
        logits, vid = eng.branch_forward(tf)
        input["logits_per_image"] = logits
        return vid, input
