"""TEST INFRASTRUCTURE ONLY (never imported by the product path): CPU restatement of the fp8 operand path of BASELINE config 5
("ViT-L/14 32+64f fp8 weights (CDNA4 fp8 MFMA) frozen spatial branch").

PARITY UNPINNED against the reference: alibaba-mmai-research/DiST has no fp8 code path (its frozen tower runs in fp32 / fp16), so there
is no golden vector, test or call site to pin to; config 5 exists only in this build's BASELINE.json.  What is pinned instead:
  * quant_rows - the per-row e4m3 quantisation - against torch's own `float8_e4m3fn` cast (OCP e4m3, round to nearest even), which this
    file uses, and the HIP kernel must reproduce bit for bit (tests/test_fp8.py);
  * gemm - fp64 product of the DEQUANTISED operands times the row scales, i.e. exactly what the block-scaled MFMA path computes up to
    fp32 accumulation order; the quantisation error itself is reported against the unquantised product in the test."""
import numpy as np
import torch


def quant_rows(x):
    """x [rows, K] (any float dtype) -> (q uint8 [rows, K] e4m3 bytes, scale fp32 [rows]); amax / 448 scaling, all-zero rows keep scale 1."""
    xf = torch.as_tensor(x).float()
    amax = xf.abs().amax(dim=1)
    inv = torch.where(amax > 0, torch.tensor(448.0) / amax, torch.ones_like(amax))          # fp32 division, as the kernel
    scale = torch.where(amax > 0, amax / torch.tensor(448.0), torch.ones_like(amax))
    q = (xf * inv[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequant(q, scale):
    return torch.as_tensor(q).view(torch.float8_e4m3fn).double() * torch.as_tensor(scale).double()[:, None]


def gemm(qa, sa, qb, sb):
    """C[m][n] = sa[m] * sb[n] * sum_k e4m3(qa[m][k]) * e4m3(qb[n][k]) in fp64"""
    return dequant(qa, sa) @ dequant(qb, sb).t()
