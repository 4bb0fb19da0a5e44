"""Oracle: the NSF (f0-conditioned) branch of NeuralSeq's HiFi-GAN, functional CPU fp32.  TEST INFRASTRUCTURE ONLY.
Groundwork for SURVEY 8f / N1 -- there is no HIP implementation of this branch yet.

Restates:
  /root/reference/NeuralSeq/modules/hifigan/hifigan.py:104-169   (HifiGanGenerator with use_pitch_embed: f0 upsample,
      m_source, per-stage noise_convs added after each ups[i])
  /root/reference/NeuralSeq/modules/parallel_wavegan/models/source.py:311-436 (SineGen, normal branch),
      :484-539 (SourceModuleHnNSF: tanh(Linear(sines)))
The reference draws two random tensors inside SineGen.forward (source.py:355-358 `torch.rand` for the initial phase of
the overtones, :425 `torch.randn_like` for the additive noise); they are INPUTS here (`rand_ini`, `noise`), generated
by the caller in the reference's order from the same seed, so the comparison is exact and a device implementation can
take them through the ABI.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .vocoder import LRELU_SLOPE, resblock

HARMONIC_NUM = 8            # hifigan.py:112
SINE_AMP = 0.1              # source.py:493 default
NOISE_STD = 0.003


def sine_source(f0_up, sampling_rate, rand_ini, noise, harmonic_num=HARMONIC_NUM, sine_amp=SINE_AMP,
                noise_std=NOISE_STD, voiced_threshold=0.0):
    """SineGen.forward (source.py:399-436).  f0_up [B, L, 1] (Hz, 0 = unvoiced), rand_ini [B, H+1] (column 0 ignored:
    the fundamental starts at phase 0), noise [B, L, H+1] ~ N(0, 1).  Returns (sine_waves [B, L, H+1], uv [B, L, 1])."""
    B, L, _ = f0_up.shape
    mult = torch.arange(1, harmonic_num + 2, dtype=f0_up.dtype)             # fundamental + overtones (:409-413)
    f0_buf = f0_up * mult                                                    # [B, L, H+1]
    rad = (f0_buf / sampling_rate) % 1                                       # :349
    ini = rand_ini.clone()
    ini[:, 0] = 0                                                            # :357
    rad[:, 0, :] = rad[:, 0, :] + ini                                        # :358
    tmp_over_one = torch.cumsum(rad, 1) % 1                                  # :369
    over_idx = (tmp_over_one[:, 1:, :] - tmp_over_one[:, :-1, :]) < 0
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over_idx * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * sine_amp    # :375-376, :416
    uv = (f0_up > voiced_threshold).to(f0_up.dtype)                          # :340-344
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3                     # :424
    return sines * uv + noise_amp * noise, uv                                # :425-429


def harmonic_source(sd, f0, hop, sampling_rate, rand_ini, noise):
    """hifigan.py:146-149: nearest upsample of f0 by the hop, SourceModuleHnNSF -> har_source [B, 1, T*hop]."""
    f0_up = F.interpolate(f0[:, None], scale_factor=hop, mode="nearest").transpose(1, 2)     # torch.nn.Upsample default
    sine_wavs, uv = sine_source(f0_up, sampling_rate, rand_ini, noise)
    merged = torch.tanh(F.linear(sine_wavs, sd["m_source.l_linear.weight"], sd["m_source.l_linear.bias"]))   # source.py:533
    return merged.transpose(1, 2)


def hifigan_nsf_forward(sd, cfg, mel, f0, rand_ini, noise):
    """hifigan.py:144-169 with f0 given.  mel [B,80,T], f0 [B,T] -> wav [B,1,T*hop].  `sd` has folded weights."""
    rates, ksz = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    hop = int(np.prod(rates))
    har = harmonic_source(sd, f0, hop, cfg["sampling_rate"], rand_ini, noise)
    nk = len(cfg["resblock_kernel_sizes"])
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i + 1 < len(rates):                                               # hifigan.py:124-129
            s = int(np.prod(rates[i + 1:]))
            xs = F.conv1d(har, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"], stride=s, padding=s // 2)
        else:
            xs = F.conv1d(har, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"])
        x = x + xs                                                           # :155-157
        acc = None
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = resblock(sd, cfg, f"resblocks.{i * nk + j}.", x, rk, rd)
            acc = r if acc is None else acc + r
        x = acc / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def draw_source_noise(seed, B, L, harmonic_num=HARMONIC_NUM):
    """The two draws of SineGen.forward in the reference's order under torch.manual_seed(seed) on CPU."""
    torch.manual_seed(seed)
    rand_ini = torch.rand(B, harmonic_num + 1)
    noise = torch.randn(B, L, harmonic_num + 1)
    return rand_ini, noise
