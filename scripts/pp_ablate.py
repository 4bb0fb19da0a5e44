"""Where does a tick of the ping-pong engine go?  Kernel-only timings of one conv with parts of the K loop switched off in the
TUNE instantiation (MAA_PP_DBG bits: 1 no MFMAs, 2 no copies after the prologue, 4 no fragment reads, 8 no waits).
python scripts/pp_ablate.py"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [("640->640 @5x39", 16, 5, 39, 640, 640), ("320->320 @10x78", 16, 10, 78, 320, 320)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from audiogpt_amd.backend import Context
    ctx = Context("cuda:0", precision="bf16x3")
    print(" ".join("%7.1f" % (ctx.op_bench_conv(B, H, W, ci, co, 9, True, 30) * 1e3) for _, B, H, W, ci, co in SHAPES), flush=True)
    sys.exit(0)
print("columns (us per launch):", ", ".join(s[0] for s in SHAPES))
VARIANTS = ((None, "product instantiation"), (0, "TUNE, everything on"), (1, "no MFMAs"), (2, "no copies"), (4, "no fragment reads"),
            (8, "no vmcnt wait"), (3, "no MFMAs, no copies"), (5, "no MFMAs, no reads"), (6, "no copies, no reads"), (7, "barriers only"),
            (16, "reads before copies"), (32, "fragments waited before barrier"), (48, "both (the v2 order)"), (64, "setprio 1 memory phase"),
            (128, "setprio 1 matrix phase"))
if len(sys.argv) > 1 and sys.argv[1] == "short":
    VARIANTS = tuple(v for v in VARIANTS if v[0] in (None, 0, 1, 2, 16, 32, 48, 64, 128))
for pp in ("128,1", "128,3"):
    for dbg, what in VARIANTS:
        e = dict(os.environ)
        e["MAA_PP"] = pp
        if dbg is not None:
            e["MAA_PP_DBG"] = str(dbg)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print("MAA_PP=%-6s dbg %-4s %-26s %s" % (pp, dbg, what, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
