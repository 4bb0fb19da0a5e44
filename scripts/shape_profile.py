"""Per-shape timing of one T2A batch (eager, hipEvent per launch): where the igemm time goes.
    python scripts/shape_profile.py [ddim_steps] > gpurun_out/shapes.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiogpt_amd.pipeline import MakeAnAudio  # noqa: E402
from bench import synth_conditioning, LATENT, CFG_SCALE  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 5
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
n = 8
pipe = MakeAnAudio("cuda:0", precision=PREC)
x_T = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float().cuda()
c = synth_conditioning(n, 1234).cuda()
uc = synth_conditioning(1, 1235).cuda().expand(n, -1, -1).contiguous()
pipe.generate(x_T, c, uc, CFG_SCALE, 2, use_graph=False)
for stage in ("unet", "vae", "vocoder"):
    pipe.ctx.prof_begin(detail=True)
    if stage == "unet":
        z = pipe.sample_latents(x_T, c, uc, CFG_SCALE, S, use_graph=False)
    elif stage == "vae":
        spec = pipe.decode(z)
    else:
        pipe.vocode(spec)
    rows = pipe.ctx.prof_end()
    tot = sum(r["ms"] for r in rows.values())
    div = S if stage == "unet" else 1
    print("== %s: %.3f ms per %s, %d rows" % (stage, tot / div, "DDIM step (B=16)" if stage == "unet" else "batch of 8", len(rows)))
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:45]:
        print("%-44s n %5d  ms/launch %8.3f  total %9.3f (%5.1f%%)  TF/s %7.2f  GB/s %8.1f" % (
            k, v["launches"] // div, v["ms"] / v["launches"], v["ms"] / div, 100 * v["ms"] / tot,
            v["flops"] / max(v["ms"], 1e-9) / 1e9, v["bytes"] / max(v["ms"], 1e-9) / 1e6))
