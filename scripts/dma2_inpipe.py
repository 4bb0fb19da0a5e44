"""In-pipeline A/B of engine policies: the benchmark's T2A batch (8 latents + CFG), 20 DDIM steps + VAE + HiFi-GAN, hipGraph
replay, one process (MAA_DMA2* are read by the library on every launch, and sample() captures its graph anew on every call).

    python scripts/dma2_inpipe.py "NAME=ENV1=v1 ENV2=v2" ...        -> ms per 20-step batch, best of 3
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiogpt_amd.pipeline import MakeAnAudio  # noqa: E402
from bench import CFG_SCALE, LATENT, synth_conditioning  # noqa: E402

pipe = MakeAnAudio("cuda:0", precision="bf16x3")
n, S = 8, 20
x_T = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float().cuda()
c = synth_conditioning(n, 1234).cuda()
uc = synth_conditioning(1, 1235).cuda().expand(n, -1, -1).contiguous()
KEYS = set()
for spec in sys.argv[1:]:
    name, _, envs = spec.partition("=")
    kv = dict(e.split("=", 1) for e in envs.split()) if envs else {}
    KEYS |= set(kv)
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(kv)
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.generate(x_T, c, uc, CFG_SCALE, S)
        torch.cuda.synchronize()
        if it:
            best = min(best, time.perf_counter() - t0)
    print("%-28s %8.2f ms   %s" % (name, best * 1e3, envs), flush=True)
