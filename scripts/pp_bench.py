"""Kernel-only timings (us per launch, split-K reduce included) of the UNet's 3x3 convolutions at batch 16 on the halo-staged
ping-pong engine (MAA_PP = tile width, K slices) against the second LDS-DMA engine (MAA_PP=off).
python scripts/pp_bench.py [precision]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
SHAPES = [("320->320 @10x78", 16, 10, 78, 320, 320), ("640->320 @10x78", 16, 10, 78, 640, 320), ("960->320 @10x78", 16, 10, 78, 960, 320),
          ("320->640 @5x39", 16, 5, 39, 320, 640), ("640->640 @5x39", 16, 5, 39, 640, 640), ("1280->640 @5x39", 16, 5, 39, 1280, 640),
          ("640->640 @5x39 b2", 2, 5, 39, 640, 640), ("320->320 @10x106", 8, 10, 106, 320, 320)]
LIN = [("qkv 320->960 M12480", 16, 10, 78, 320, 960), ("out 320->320 M12480", 16, 10, 78, 320, 320), ("geglu 320->2560 M12480", 16, 10, 78, 320, 2560),
       ("ff2 1280->320 M12480", 16, 10, 78, 1280, 320), ("qkv 640->1920 M3120", 16, 5, 39, 640, 1920), ("out 640->640 M3120", 16, 5, 39, 640, 640),
       ("geglu 640->5120 M3120", 16, 5, 39, 640, 5120), ("ff2 2560->640 M3120", 16, 5, 39, 2560, 640)]
if len(sys.argv) > 2 and sys.argv[2] == "child1":
    from audiogpt_amd.backend import Context
    ctx = Context("cuda:0", precision=prec)
    out = []
    for name, B, H, W, ci, co in LIN:
        try:
            out.append("%7.1f" % (ctx.op_bench_conv(B, H, W, ci, co, 1, True, 30) * 1e3))
        except Exception as e:
            out.append("   fail")
    print(" ".join(out), flush=True)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "child":
    from audiogpt_amd.backend import Context
    ctx = Context("cuda:0", precision=prec)
    out = []
    for name, B, H, W, ci, co in SHAPES:
        try:
            out.append("%7.1f" % (ctx.op_bench_conv(B, H, W, ci, co, 9, True, 30) * 1e3))
        except Exception as e:
            out.append("   fail")
    print(" ".join(out), flush=True)
    sys.exit(0)
print("columns (us per launch):")
for i, s in enumerate(SHAPES):
    B, H, W, ci, co = s[1:]
    print("  [%d] %-20s %6.2f GFLOP  MFMA floor %.1f us" % (i, s[0], 2.0 * B * H * W * co * 9 * ci / 1e9, 3 * 2.0 * B * H * W * co * 9 * ci / 2.5e15 * 1e6))
for tag in ["off", "", "128,1", "128,2", "128,3", "128,4", "128,5", "160,1", "160,2", "160,3", "160,4", "160,5"]:
    e = dict(os.environ)
    e["MAA_PP"] = tag
    r = subprocess.run([sys.executable, __file__, prec, "child"], env=e, capture_output=True, text=True)
    print("%-8s %s" % (tag or "default", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:]), flush=True)
print("1x1 columns (us per launch; the geglu rows are timed as plain linears of the packed width):")
for i, s in enumerate(LIN):
    B, H, W, ci, co = s[1:]
    print("  [%d] %-24s %6.2f GFLOP  MFMA floor %.1f us" % (i, s[0], 2.0 * B * H * W * co * ci / 1e9, 3 * 2.0 * B * H * W * co * ci / 2.5e15 * 1e6))
for tag in ["off", "", "128,1", "128,2", "160,1", "160,2"]:
    e = dict(os.environ)
    e["MAA_PP1"] = tag
    r = subprocess.run([sys.executable, __file__, prec, "child1"], env=e, capture_output=True, text=True)
    print("%-8s %s" % (tag or "default", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:]), flush=True)
