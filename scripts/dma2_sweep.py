"""Kernel-only timings of the UNet's main contractions under each configuration of the wide-tile / split-K engine
(igemm_dma2.hip) next to the round-1 engines.  One process; MAA_DMA2 is re-read by the library on every launch.

    python scripts/dma2_sweep.py [quick|full|ablate]      (bf16x3, us per launch, 30 launches each)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "full"

# name, B, H, W, Cin, Cout, taps          (batch 16 = 8 latents + CFG, the benchmark's UNet batch)
SHAPES = [("c320-320@10x78", 16, 10, 78, 320, 320, 9), ("c640-320@10x78", 16, 10, 78, 640, 320, 9),
          ("c960-320@10x78", 16, 10, 78, 960, 320, 9), ("c640-640@10x78", 16, 10, 78, 640, 640, 9),
          ("c640-640@5x39", 16, 5, 39, 640, 640, 9), ("c1280-640@5x39", 16, 5, 39, 1280, 640, 9),
          ("c320-640@5x39", 16, 5, 39, 320, 640, 9),
          ("l320-320 M12480", 16, 10, 78, 320, 320, 1), ("l1280-320 M12480", 16, 10, 78, 1280, 320, 1),
          ("l320-960 M12480", 16, 10, 78, 320, 960, 1),
          ("l640-640 M3120", 16, 5, 39, 640, 640, 1), ("l2560-640 M3120", 16, 5, 39, 2560, 640, 1),
          ("l640-1920 M3120", 16, 5, 39, 640, 1920, 1)]

if mode == "quick":
    SHAPES = [SHAPES[0], SHAPES[4], SHAPES[8], SHAPES[11]]

CONFIGS = [("r1 engines", "off")]
# (tile cfg, LDS stages, in-wave pipelining)
VAR = ((0, 2, 0), (0, 4, 1), (1, 3, 1), (2, 2, 0), (3, 2, 0), (3, 4, 0), (3, 4, 1), (4, 3, 0), (4, 4, 1))
for cfg, ns, pipe in VAR:
    for S in (1, 2, 4):
        CONFIGS.append(("t%d ns%d p%d S%d" % (cfg, ns, pipe, S), "%d,%d,%d,%d" % (cfg, ns, pipe, S)))
if mode == "quick":
    CONFIGS = [c for c in CONFIGS if c[1] in ("off", "0,4,1,2", "2,2,0,2", "3,4,0,1", "3,2,0,1", "4,3,0,1")]
if mode == "ablate":
    CONFIGS = [("r1 engines", "off"), ("t0 ns4 p1 S2", "0,4,1,2"), ("t3 ns4 S1", "3,4,0,1"), ("t2 ns2 S2", "2,2,0,2")]

from audiogpt_amd.backend import Context  # noqa: E402

ctx = Context("cuda:0", precision="bf16x3")
flops = [2.0 * B * H * W * ci * co * t for _, B, H, W, ci, co, t in SHAPES]
print("us per launch | TFLOP/s (algorithmic 2MNK); columns:")
for i, s in enumerate(SHAPES):
    print("  [%d] %s  (%.1f GFLOP)" % (i, s[0], flops[i] / 1e9))
# ablation masks of the new engine (MAA_DBG, read per launch): 1 = no fragment reads / MFMAs, 2 = no tile copies
dbgs = [0, 1, 2, 3] if mode == "ablate" else [0]
best = [(1e9, "")] * len(SHAPES)
for dbg in dbgs:
    os.environ["MAA_DBG"] = str(dbg)
    if mode == "ablate":
        print("== MAA_DBG=%d" % dbg)
    for tag, env in CONFIGS:
        if dbg and env == "off":
            continue
        os.environ["MAA_DMA2"] = env
        row = []
        for i, (name, B, H, W, ci, co, taps) in enumerate(SHAPES):
            try:
                us = ctx.op_bench_conv(B, H, W, ci, co, taps, True, 30) * 1e3
            except Exception as e:  # a configuration the library refuses
                us = float("nan")
                sys.stderr.write("%s %s: %s\n" % (tag, name, str(e)[:200]))
            row.append(us)
            if us < best[i][0]:
                best[i] = (us, tag)
        print("%-14s " % tag + " ".join("%7.1f" % u for u in row), flush=True)
        print("%-14s " % "" + " ".join("%7.0f" % (flops[i] / (row[i] * 1e-6) / 1e12) for i in range(len(row))), flush=True)
os.environ["MAA_DBG"] = "0"
print("best per shape:")
for i, s in enumerate(SHAPES):
    print("  %-20s %8.1f us  %6.0f TFLOP/s  %s" % (s[0], best[i][0], flops[i] / (best[i][0] * 1e-6) / 1e12, best[i][1]))
