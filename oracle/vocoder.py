"""Oracle: HiFi-GAN / BigVGAN MRF generators, functional CPU fp32.  TEST INFRASTRUCTURE ONLY.

Restates:
  /root/reference/NeuralSeq/modules/hifigan/hifigan.py:26-91, 104-178  (ResBlock1 / ResBlock2, HifiGanGenerator, f0=None)
  /root/reference/text_to_audio/Make_An_Audio/vocoder/hifigan/modules.py:22-83, 86-136 (Generator; same graph)
  /root/reference/text_to_audio/Make_An_Audio/vocoder/bigvgan/models.py:30-132, 133-203 (AMPBlock1 / AMPBlock2, BigVGAN)
  .../vocoder/bigvgan/activations.py:62-119 (Snake / SnakeBeta)
  .../vocoder/bigvgan/alias_free_torch/{act.py:8-27, filter.py:28-94, resample.py:10-49}
State-dict keys follow the reference generators; weight-norm pairs (weight_g / weight_v) are folded
the way torch.nn.utils.remove_weight_norm does.
"""
import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1


def fold_weight_norm(sd):
    """`w = g * v / ||v||`, norm over all dims but 0 (torch weight_norm default dim=0; for
    ConvTranspose1d dim 0 is the in-channel axis -- hifigan.py:171-178)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            g = v
            vv = sd[base + ".weight_v"]
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (vv.dim() - 1)))
            out[base + ".weight"] = vv * (g / norm)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = v
    return out


def _pad(k, d=1):
    return int((k * d - d) / 2)


def _resblock1(sd, p, x, k, dil):
    """hifigan.py:54-61."""
    for j, d in enumerate(dil):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[p + f"convs1.{j}.weight"], sd[p + f"convs1.{j}.bias"], padding=_pad(k, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[p + f"convs2.{j}.weight"], sd[p + f"convs2.{j}.bias"], padding=_pad(k, 1))
        x = xt + x
    return x


def _resblock2(sd, p, x, k, dil):
    """hifigan.py:83-88 (`resblock: "2"`): one dilated conv per residual step."""
    for j, d in enumerate(dil):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[p + f"convs.{j}.weight"], sd[p + f"convs.{j}.bias"], padding=_pad(k, d), dilation=d)
        x = xt + x
    return x


def resblock(sd, cfg, p, x, k, dil):
    """hifigan.py:119 / modules.py:93: `ResBlock1 if h.resblock == '1' else ResBlock2`."""
    return (_resblock1 if str(cfg.get("resblock", "1")) == "1" else _resblock2)(sd, p, x, k, dil)


def hifigan_forward(sd, cfg, mel):
    """hifigan.py:144-169 with f0=None.  mel [B,80,T] -> wav [B,1,T*hop].  `sd` has folded weights."""
    nk = len(cfg["resblock_kernel_sizes"])
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r = resblock(sd, cfg, f"resblocks.{i * nk + j}.", x, rk, rd)
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)          # slope 0.01 (hifigan.py:165)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


# ------------------------------------------------------------------------------ BigVGAN
def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """filter.py:28-57 -> tensor [kernel_size]."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if even:
        time = torch.arange(-half_size, half_size) + 0.5
    else:
        time = torch.arange(kernel_size) - half_size
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return filt / filt.sum()


def upsample1d(x, filt, ratio=2):
    """resample.py:25-33."""
    ks = filt.numel()
    pad = ks // ratio - 1
    pad_left = pad * ratio + (ks - ratio) // 2
    pad_right = pad * ratio + (ks - ratio + 1) // 2
    C = x.shape[1]
    x = F.pad(x, (pad, pad), mode="replicate")
    x = ratio * F.conv_transpose1d(x, filt.view(1, 1, -1).expand(C, -1, -1), stride=ratio, groups=C)
    return x[..., pad_left:-pad_right]


def downsample1d(x, filt, ratio=2):
    """resample.py:46-49 -> filter.py:86-94 (replicate pad (k/2-1, k/2), stride ratio)."""
    ks = filt.numel()
    C = x.shape[1]
    x = F.pad(x, (ks // 2 - int(ks % 2 == 0), ks // 2), mode="replicate")
    return F.conv1d(x, filt.view(1, 1, -1).expand(C, -1, -1), stride=ratio, groups=C)


def snake(x, alpha, beta, logscale):
    """activations.py:107-119 (SnakeBeta); Snake (:46-59) is beta := alpha."""
    a = alpha[None, :, None]
    b = beta[None, :, None]
    if logscale:
        a = torch.exp(a)
        b = torch.exp(b)
    return x + (1.0 / (b + 1e-9)) * torch.pow(torch.sin(x * a), 2)


def activation1d(sd, p, x, cfg, filt):
    """act.py:23-27: up x2 -> snake -> down x2."""
    alpha = sd[p + "act.alpha"]
    beta = sd[p + "act.beta"] if cfg["activation"] == "snakebeta" else alpha
    x = upsample1d(x, filt)
    x = snake(x, alpha, beta, cfg["snake_logscale"])
    return downsample1d(x, filt)


def _ampblock1(sd, p, x, k, dil, cfg, filt):
    """bigvgan/models.py:72-81."""
    for j, d in enumerate(dil):
        xt = activation1d(sd, p + f"activations.{2 * j}.", x, cfg, filt)
        xt = F.conv1d(xt, sd[p + f"convs1.{j}.weight"], sd[p + f"convs1.{j}.bias"], padding=_pad(k, d), dilation=d)
        xt = activation1d(sd, p + f"activations.{2 * j + 1}.", xt, cfg, filt)
        xt = F.conv1d(xt, sd[p + f"convs2.{j}.weight"], sd[p + f"convs2.{j}.bias"], padding=_pad(k, 1))
        x = xt + x
    return x


def _ampblock2(sd, p, x, k, dil, cfg, filt):
    """bigvgan/models.py:122-128 (`resblock: "2"`)."""
    for j, d in enumerate(dil):
        xt = activation1d(sd, p + f"activations.{j}.", x, cfg, filt)
        xt = F.conv1d(xt, sd[p + f"convs.{j}.weight"], sd[p + f"convs.{j}.bias"], padding=_pad(k, d), dilation=d)
        x = xt + x
    return x


def bigvgan_forward(sd, cfg, mel):
    """bigvgan/models.py:181-203.  `sd` has folded weights; ups keys are `ups.{i}.0.*`."""
    nk = len(cfg["resblock_kernel_sizes"])
    filt = kaiser_sinc_filter1d(0.25, 0.3, 12)
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, sd[f"ups.{i}.0.weight"], sd[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            amp = _ampblock1 if str(cfg.get("resblock", "1")) == "1" else _ampblock2       # models.py:146
            r = amp(sd, f"resblocks.{i * nk + j}.", x, rk, rd, cfg, filt)
            xs = r if xs is None else xs + r
        x = xs / nk
    x = activation1d(sd, "activation_post.", x, cfg, filt)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def vocoder_forward(sd, cfg, mel):
    if cfg["kind"] == "bigvgan":
        return bigvgan_forward(sd, cfg, mel)
    return hifigan_forward(sd, cfg, mel)
