"""Where the short-K linears' time goes on the 64x64 LDS-DMA engine: K scan of one [12480 x K] x [K x 320] product (and the
5x39-level twin) under the timing ablations (MAA_DBG: 1 = no fragment reads / MFMAs, 2 = no tile copies, 3 = neither: the
loop skeleton + epilogue).  MAA_DBG / MAA_DMA_NS are read once per process by that engine: run one process per setting.

    MAA_DMA2=off MAA_DBG=3 MAA_DMA_NS=2 python scripts/shortk_ablate.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiogpt_amd.backend import Context  # noqa: E402

ctx = Context("cuda:0", precision="bf16x3")
SHAPES = [(16, 10, 78, k, 320) for k in (64, 128, 320, 640, 1280)] + [(16, 5, 39, k, 640) for k in (128, 640, 2560)] + \
         [(16, 10, 78, 320, 960), (16, 5, 39, 640, 1920)]
tag = "dbg=%s ns=%s dma2=%s" % (os.environ.get("MAA_DBG", "0"), os.environ.get("MAA_DMA_NS", "-"), os.environ.get("MAA_DMA2", "-"))
row = []
for B, H, W, ci, co in SHAPES:
    row.append(ctx.op_bench_conv(B, H, W, ci, co, 1, True, 40) * 1e3)
print("%-28s " % tag + " ".join("%7.1f" % u for u in row), flush=True)
