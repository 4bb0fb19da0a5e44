"""Kernel-only conv/linear timings on the main UNet shapes, for each tile config.  python scripts/conv_bench.py [precision]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
if len(sys.argv) > 2:      # child: run all shapes under the current env
    from audiogpt_amd.backend import Context
    ctx = Context("cuda:0", precision=prec)
    shapes = [("conv 320->320 @10x78 B16", 16, 10, 78, 320, 320, 9), ("conv 640->640 @5x39 B16", 16, 5, 39, 640, 640, 9),
              ("conv 320->320 @10x78 B128", 128, 10, 78, 320, 320, 9), ("conv 640->640 @5x39 B128", 128, 5, 39, 640, 640, 9),
              ("lin 320->320 M12480", 16, 10, 78, 320, 320, 1), ("lin 640->640 M3120", 16, 5, 39, 640, 640, 1),
              ("lin 1280->320 M12480", 16, 10, 78, 1280, 320, 1), ("lin 320->320 M99840", 128, 10, 78, 320, 320, 1)]
    for name, B, H, W, ci, co, taps in shapes:
        ms = ctx.op_bench_conv(B, H, W, ci, co, taps, True, 30)
        fl = 2.0 * B * H * W * co * ci * taps
        print("  %-28s %8.1f us  %7.1f TFLOP/s" % (name, ms * 1e3, fl / ms / 1e9))
    sys.exit(0)
for env in ({}, {"MAA_FORCE_CFG": "0"}, {"MAA_FORCE_CFG": "1"}, {"MAA_FORCE_CFG": "2"}, {"MAA_NBUF": "1", "MAA_FORCE_CFG": "2"},
            {"MAA_NBUF": "1", "MAA_FORCE_CFG": "1"}):
    print("==", prec, env or "default", flush=True)
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, __file__, prec, "child"], env=e)
