"""Oracle: DiffSinger's denoiser (DiffNet) and its PLMS sampling loop, functional CPU fp32.  TEST INFRASTRUCTURE ONLY.
Groundwork for SURVEY 8f / N2 (the T2S tool's diffusion hot loop, audio-chatgpt.py:298-339) -- no HIP implementation yet.

Restates:
  /root/reference/NeuralSeq/modules/diff/net.py:31-44 (SinusoidalPosEmb), :58-81 (ResidualBlock), :84-130 (DiffNet)
  /root/reference/NeuralSeq/modules/diff/diffusion.py:68-70 (Mish)
  /root/reference/NeuralSeq/modules/diff/shallow_diffusion_tts.py:43-49 (linear_beta_schedule), :71-96 (buffers),
      :166-201 (p_sample_plms: pseudo linear multi-step with a 4-deep noise history), :262-269 (the sampling loop with
      pndm_speedup), :279-283 (norm / denorm of the mel)
"""
import math
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F


def sinusoidal_pos_emb(t, dim):
    """net.py:36-44: [sin | cos] of t * exp(-ln(1e4) * i / (dim/2 - 1))."""
    half = dim // 2
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    emb = t[:, None].float() * emb[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def mish(x):
    return x * torch.tanh(F.softplus(x))


def diffnet_forward(sd, cfg, spec, t, cond):
    """net.py:107-130.  spec [B,1,M,T], t [B] (int), cond [B,H,T] -> predicted noise [B,1,M,T]."""
    C = cfg["residual_channels"]
    x = F.relu(F.conv1d(spec[:, 0], sd["input_projection.weight"], sd["input_projection.bias"]))
    d = sinusoidal_pos_emb(t, C)
    d = F.linear(mish(F.linear(d, sd["mlp.0.weight"], sd["mlp.0.bias"])), sd["mlp.2.weight"], sd["mlp.2.bias"])
    skip = None
    for i in range(cfg["residual_layers"]):
        p = f"residual_layers.{i}."
        dil = 2 ** (i % cfg["dilation_cycle_length"])
        step = F.linear(d, sd[p + "diffusion_projection.weight"], sd[p + "diffusion_projection.bias"]).unsqueeze(-1)
        c = F.conv1d(cond, sd[p + "conditioner_projection.weight"], sd[p + "conditioner_projection.bias"])
        y = F.conv1d(x + step, sd[p + "dilated_conv.weight"], sd[p + "dilated_conv.bias"], padding=dil, dilation=dil) + c
        gate, filt = torch.chunk(y, 2, dim=1)
        y = torch.sigmoid(gate) * torch.tanh(filt)
        y = F.conv1d(y, sd[p + "output_projection.weight"], sd[p + "output_projection.bias"])
        residual, s = torch.chunk(y, 2, dim=1)
        x = (x + residual) / math.sqrt(2.0)
        skip = s if skip is None else skip + s
    x = skip / math.sqrt(cfg["residual_layers"])
    x = F.relu(F.conv1d(x, sd["skip_projection.weight"], sd["skip_projection.bias"]))
    x = F.conv1d(x, sd["output_projection.weight"], sd["output_projection.bias"])
    return x[:, None]


def alphas_cumprod(timesteps, max_beta):
    """shallow_diffusion_tts.py:43-49,82-96: linear betas in fp64, cumprod, stored as an fp32 buffer."""
    betas = np.linspace(1e-4, max_beta, timesteps)
    return torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)


def plms_x_pred(ac, x, noise_t, t, interval):
    """get_x_pred of p_sample_plms (:172-183)."""
    a_t = ac[t].reshape(-1, 1, 1, 1)
    if int(t[0]) < interval:
        a_prev = torch.ones_like(a_t)
    else:
        a_prev = ac[torch.clamp(t - interval, min=0)].reshape(-1, 1, 1, 1)
    a_t_sq, a_prev_sq = a_t.sqrt(), a_prev.sqrt()
    x_delta = (a_prev - a_t) * ((1 / (a_t_sq * (a_t_sq + a_prev_sq))) * x -
                                1 / (a_t_sq * (((1 - a_prev) * a_t).sqrt() + ((1 - a_t) * a_prev).sqrt())) * noise_t)
    return x + x_delta


def plms_sample(denoise, ac, x, cond, K_step, interval, trace=None):
    """shallow_diffusion_tts.py:262-269 + 166-201: t = K_step - interval ... 0 in steps of `interval`."""
    hist = deque(maxlen=4)
    b = x.shape[0]
    for i in reversed(range(0, K_step, interval)):
        t = torch.full((b,), i, dtype=torch.long)
        e = denoise(x, t, cond)
        if len(hist) == 0:
            x_pred = plms_x_pred(ac, x, e, t, interval)
            e_prev = denoise(x_pred, torch.clamp(t - interval, min=0), cond)
            e_prime = (e + e_prev) / 2
        elif len(hist) == 1:
            e_prime = (3 * e - hist[-1]) / 2
        elif len(hist) == 2:
            e_prime = (23 * e - 16 * hist[-1] + 5 * hist[-2]) / 12
        else:
            e_prime = (55 * e - 59 * hist[-1] + 37 * hist[-2] - 9 * hist[-3]) / 24
        x = plms_x_pred(ac, x, e_prime, t, interval)
        hist.append(e)
        if trace is not None:
            trace.append(x.clone())
    return x


def denorm_spec(x, spec_min, spec_max):
    """:282-283 on x [B,1,M,T] -> mel [B,T,M]."""
    m = x[:, 0].transpose(1, 2)
    return (m + 1) / 2 * (spec_max - spec_min) + spec_min
