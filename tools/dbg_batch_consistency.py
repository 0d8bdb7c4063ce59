"""Which intermediate tensor of the b=32 run first departs from the b=2 run (rows of clips 0-1)?"""
import sys, os
os.environ.setdefault("DIST_AMD_KEEP_MID", "1")   # M' of every layer stays readable ("mid.i"): the fused IntegrationNetwork forward forms it on chip otherwise
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_engine_gpu import build
g, eng, sd, video, text, tgt = build("b16_8+16f", 32, torch.bfloat16)
loss, logits = eng.forward_backward(video, text, tgt)
g2, eng2, _, _, _, _ = build("b16_8+16f", 2, torch.bfloat16)
loss2, logits2 = eng2.forward_backward(video[:2].contiguous(), text, tgt[:2].contiguous())
torch.cuda.synchronize()
L = eng.cfg.layers if hasattr(eng, "cfg") else 12
names = ["stem"] + [f"{k}.{i}" for i in range(12) for k in ("feat", "tn_out", "mid", "int_out", "x_temporal")]
for n in names:
    try:
        a, b = eng.debug(n), eng2.debug(n)
    except Exception as e:
        print(n, "unavailable", str(e)[:60]); continue
    r = b.shape[0]
    d = (a[:r].float() - b.float()).abs()
    bad = (d > 0).nonzero()
    print(f"{n:14s} rows {r} max diff {float(d.max()):.5f} mismatched {len(bad)}" + (f" first at {bad[0].tolist()} rows-with-diff {sorted(set(bad[:,0].tolist()))[:12]}" if len(bad) else ""))
print("logits diff", float((logits[:2].float() - logits2.float()).abs().max()))
