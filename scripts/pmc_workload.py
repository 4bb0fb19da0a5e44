"""Small eager workload for PMC collection: a few T2A DDIM steps (B=8 latents + CFG) in the given precision."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiogpt_amd.pipeline import MakeAnAudio  # noqa: E402
from bench import synth_conditioning, LATENT, CFG_SCALE  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pipe = MakeAnAudio("cuda:0", precision=prec)
pipe.ctx.set_cfg_split(False)      # the arrangement of bench.py's roofline pass (the headline's replicas): a CFG step on ONE stream, so the
                                   # launches per DDIM step match and attach_traffic accepts the tables (the library default is two lanes)
n = 8
x_T = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float().cuda()
c = synth_conditioning(n, 1234).cuda()
uc = synth_conditioning(1, 1235).cuda().expand(n, -1, -1).contiguous()
z = pipe.sample_latents(x_T, c, uc, CFG_SCALE, S, use_graph=False)
torch.cuda.synchronize()
