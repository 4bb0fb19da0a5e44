"""Is there latent concurrency to harvest?  The benchmark batch (8 latents + CFG, 20 DDIM steps + decode) as ONE pipeline vs
the same work as TWO pipelines of 4 latents on two library streams driven by two host threads (ctypes drops the GIL).
If two half-batches side by side beat one full batch, splitting the UNet batch over two graph branches would pay.

    python scripts/dual_stream_probe.py
"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiogpt_amd.pipeline import MakeAnAudio  # noqa: E402
from bench import CFG_SCALE, LATENT, synth_conditioning  # noqa: E402

S = 20
x_T = torch.from_numpy(np.random.RandomState(55).randn(8, *LATENT)).float().cuda()
c = synth_conditioning(8, 1234).cuda()
uc = synth_conditioning(1, 1235).cuda().expand(8, -1, -1).contiguous()
pipes = [MakeAnAudio("cuda:0", precision="bf16x3") for _ in range(2)]


def run(pipe, lo, hi):
    pipe.generate(x_T[lo:hi], c[lo:hi], uc[lo:hi], CFG_SCALE, S)


def timed(jobs, reps=4):
    best = 1e9
    for it in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=run, args=j) for j in jobs]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        if it:
            best = min(best, time.perf_counter() - t0)
    return best * 1e3


print("one pipeline, batch 8                 %8.2f ms" % timed([(pipes[0], 0, 8)]))
print("one pipeline, batch 4                 %8.2f ms" % timed([(pipes[0], 0, 4)]))
print("two pipelines, batch 4 + 4 (threads)  %8.2f ms" % timed([(pipes[0], 0, 4), (pipes[1], 4, 8)]))
print("two pipelines, batch 8 + 8 (threads)  %8.2f ms   (twice the work)" % timed([(pipes[0], 0, 8), (pipes[1], 0, 8)]))
print("one pipeline, batch 8 again           %8.2f ms" % timed([(pipes[0], 0, 8)]))
