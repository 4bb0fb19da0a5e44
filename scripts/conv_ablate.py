"""Ablation timings of the bf16x3 conv kernel (MAA_DBG: 1 no MFMA phase, 2 no tile loads, 4 no LDS stores).
python scripts/conv_ablate.py"""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
print("rows: dbg mask; columns as in conv_bench.py", flush=True)
for cfg, masks in (("2", ("0", "1", "2", "4", "6", "5", "3", "7")), ("0", ("0", "1", "2", "6"))):
    for dbg in masks:
        e = dict(os.environ, MAA_FORCE_CFG=cfg, MAA_DBG=dbg)
        r = subprocess.run([sys.executable, os.path.join(here, "conv_bench.py"), "bf16x3", "child"], env=e,
                           capture_output=True, text=True)
        print("cfg%s dbg%s  %s" % (cfg, dbg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]), flush=True)
