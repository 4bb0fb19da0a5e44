"""Workload constants and synthetic inputs shared by the benchmark's parts (BASELINE.json configs[1] / [2])."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audiogpt_amd import config as C            # noqa: E402
from audiogpt_amd import weights as WT          # noqa: E402

# MI355X_MICROARCH.md: dense MFMA peaks.  The bf16x3 mode issues 3 bf16 MFMAs per algorithmic multiply-add
# (hi*hi + hi*lo + lo*hi), so its algorithmic ceiling is a third of the bf16 MFMA peak.
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}
MFMA_PER_FLOP = {"f32": 1, "bf16x3": 3, "bf16": 1}
CLIP_FRAMES = 624
LATENT = (4, 10, 78)
DDIM_STEPS = 100
CFG_SCALE = 1.5
PROMPTS_PER_GPU = 8


def synth_conditioning(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.layer_norm(torch.randn(n, 77, 1024, generator=g), (1024,))

HIFIGAN64 = dict(B=64, T=1024, seed=7)


def hifigan64_mel(B=HIFIGAN64["B"], T=HIFIGAN64["T"], seed=HIFIGAN64["seed"]):
    """BASELINE.md section 2, config 3 (the same formula as tests/golden/make_golden.py hifigan_case)."""
    g = torch.Generator().manual_seed(seed)
    return torch.clamp(torch.randn(B, 80, T, generator=g) * 1.5 - 2.25, -6.0, 1.5)

def _t2a_inputs(n, dev):
    x_T = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float().to(dev)
    return x_T, synth_conditioning(n, 1234).to(dev), synth_conditioning(1, 1235).to(dev).expand(n, -1, -1).contiguous()


def _timed(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k, out
