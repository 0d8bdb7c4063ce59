import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_fp8 import _engine, _rel
from dist_oracle import Oracle
from dist_amd import synth
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 15
g, eng, sd, video, text, tgt = _engine("b16_8+16f", 2, mask)
eng.forward_backward(video, text, tgt)
feats = [eng.debug(f"feat.{i}").clone() for i in range(g.layers)]
g0, eng0, _, _, _, _ = _engine("b16_8+16f", 2, 0)
eng0.forward_backward(video, text, tgt)
feats0 = [eng0.debug(f"feat.{i}").clone() for i in range(g.layers)]
v = torch.from_numpy(synth.video(g, 2))
o = Oracle(g, sd, dtype=torch.float32, bf16=True, vit_fp8=mask); ref = o.vit(o.patchify(v))
o16 = Oracle(g, sd, dtype=torch.float32, bf16=True); ref16 = o16.vit(o16.patchify(v))
for i in range(g.layers):
    print(f"layer {i:2d}: eng8-orc8 {_rel(feats[i], ref[i]):.4f}  eng16-orc16 {_rel(feats0[i], ref16[i]):.4f}  orc8-orc16 {_rel(ref[i], ref16[i]):.4f}  eng8-eng16 {_rel(feats[i], feats0[i]):.4f}  eng8-orc16 {_rel(feats[i], ref16[i]):.4f}")
