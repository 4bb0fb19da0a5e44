"""Kernel-only timings of the fused attention on the UNet's shapes (batch 16 = 8 prompts + CFG), from the library's own
per-launch events.  python scripts/attn_bench.py [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiogpt_amd.backend import Context  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
ctx = Context("cuda:0", precision=prec)
SHAPES = [("self  d40 780x780 (10x78)", 16, 8, 40, 780, 780), ("cross d40 780x77", 16, 8, 40, 780, 77),
          ("self  d80 195x195 (5x39)", 16, 8, 80, 195, 195), ("cross d80 195x77", 16, 8, 80, 195, 77),
          ("self  d32 780x780 (I2A)", 16, 8, 32, 780, 780), ("self  d64 257x257 (ViT-H)", 8, 16, 64, 257, 257)]
g = torch.Generator().manual_seed(0)
for name, B, h, d, nq, nk in SHAPES:
    q = torch.randn(B, nq, h * d, generator=g).cuda()
    k = torch.randn(B, nk, h * d, generator=g).cuda()
    v = torch.randn(B, nk, h * d, generator=g).cuda()
    for _ in range(3):
        ctx.op_attention(q, k, v, h, d ** -0.5)
    ctx.prof_begin()
    for _ in range(20):
        ctx.op_attention(q, k, v, h, d ** -0.5)
    rows = ctx.prof_end()
    r = rows.get("flash_attention") or next(iter(rows.values()))
    fl = 4.0 * B * h * nq * nk * d
    print("%-28s %8.1f us   %6.1f TFLOP/s algorithmic" % (name, 1e3 * r["ms"] / r["launches"], fl / (r["ms"] / r["launches"] * 1e-3) / 1e12), flush=True)
