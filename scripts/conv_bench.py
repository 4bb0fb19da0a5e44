"""Kernel-only conv/linear timings on the main UNet shapes for each (tile config, LDS-DMA stages).
python scripts/conv_bench.py [precision] [sweep|default]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
mode = sys.argv[2] if len(sys.argv) > 2 else "sweep"
SHAPES = [("conv 320->320 @10x78", 16, 10, 78, 320, 320, 9), ("conv 640->320 @10x78", 16, 10, 78, 640, 320, 9),
          ("conv 640->640 @5x39", 16, 5, 39, 640, 640, 9), ("conv 1280->640 @5x39", 16, 5, 39, 1280, 640, 9),
          ("conv 1280->1280 @3x20", 16, 3, 20, 1280, 1280, 9), ("conv 2560->1280 @3x20", 16, 3, 20, 2560, 1280, 9),
          ("lin 320->320 M12480", 16, 10, 78, 320, 320, 1), ("lin 1280->320 M12480", 16, 10, 78, 1280, 320, 1),
          ("lin 640->640 M3120", 16, 5, 39, 640, 640, 1), ("lin 2560->640 M3120", 16, 5, 39, 2560, 640, 1),
          ("lin 5120->1280 M960", 16, 3, 20, 5120, 1280, 1)]
if mode == "child":      # run all shapes under the current env
    from audiogpt_amd.backend import Context
    ctx = Context("cuda:0", precision=prec)
    out = []
    for name, B, H, W, ci, co, taps in SHAPES:
        ms = ctx.op_bench_conv(B, H, W, ci, co, taps, True, 30)
        out.append("%7.1f" % (ms * 1e3))
    print(" ".join(out), flush=True)
    sys.exit(0)
print("columns (us per launch):")
for i, s in enumerate(SHAPES):
    print("  [%d] %s" % (i, s[0]))
combos = [{}]
if mode == "sweep":
    combos += [{"MAA_NO_DMA": "1"}] + [{"MAA_DMA_NS_LOW": str(ns)} for ns in (2, 3, 4)]      # (MAA_FORCE_CFG / MAA_DMA_NS were retired in round 4)
for env in combos:
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, __file__, prec, "child"], env=e, capture_output=True, text=True)
    tag = ("regs" if "MAA_NO_DMA" in env else "NS_LOW%s" % env.get("MAA_DMA_NS_LOW", "-")) if env else "default "
    print("%-10s %s" % (tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
