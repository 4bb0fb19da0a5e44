"""Oracle: mel VAE (AutoencoderKL decoder / encoder), functional CPU fp32.  TEST INFRASTRUCTURE ONLY.

Restates (relative to /root/reference/text_to_audio/Make_An_Audio):
  ldm/modules/diffusionmodules/model.py:33-141 (swish, Normalize eps 1e-6, Upsample, Downsample, ResnetBlock)
  ldm/modules/diffusionmodules/model.py:150-202 (AttnBlock), :368-459 (Encoder), :462-568 (Decoder)
  ldm/models/autoencoder.py:345-354 (quant_conv / post_quant_conv around Encoder / Decoder)
  ldm/modules/distributions/distributions.py:24-37 (DiagonalGaussianDistribution)
State-dict keys are relative to `first_stage_model.`.
"""
import torch
import torch.nn.functional as F


def _swish(x):
    return x * torch.sigmoid(x)


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + "weight"], sd[p + "bias"], 1e-6)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=stride, padding=padding)


def resnet_block(sd, p, x):
    """model.py:121-141 with temb=None."""
    h = _conv(sd, p + "conv1.", _swish(_gn(sd, p + "norm1.", x)))
    h = _conv(sd, p + "conv2.", _swish(_gn(sd, p + "norm2.", h)))
    if (p + "nin_shortcut.weight") in sd:
        x = _conv(sd, p + "nin_shortcut.", x, padding=0)
    return x + h


def attn_block(sd, p, x):
    """model.py:178-202: single head, scale C^-1/2, softmax over keys."""
    h = _gn(sd, p + "norm.", x)
    q = _conv(sd, p + "q.", h, padding=0)
    k = _conv(sd, p + "k.", h, padding=0)
    v = _conv(sd, p + "v.", h, padding=0)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.bmm(q, k) * (int(c) ** (-0.5))
    w = F.softmax(w, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    h = _conv(sd, p + "proj_out.", h, padding=0)
    return x + h


def _levels(dd):
    return len(dd["ch_mult"])


def decoder_forward(sd, dd, z, prefix="decoder."):
    """model.py:535-568."""
    p = prefix
    nres = _levels(dd)
    curr_res = dd["resolution"] // 2 ** (nres - 1)
    h = _conv(sd, p + "conv_in.", z)
    h = resnet_block(sd, p + "mid.block_1.", h)
    h = attn_block(sd, p + "mid.attn_1.", h)
    h = resnet_block(sd, p + "mid.block_2.", h)
    for lvl in reversed(range(nres)):
        for ib in range(dd["num_res_blocks"] + 1):
            h = resnet_block(sd, p + f"up.{lvl}.block.{ib}.", h)
            if curr_res in dd["attn_resolutions"]:
                h = attn_block(sd, p + f"up.{lvl}.attn.{ib}.", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, p + f"up.{lvl}.upsample.conv.", h)
            curr_res *= 2
    h = _swish(_gn(sd, p + "norm_out.", h))
    return _conv(sd, p + "conv_out.", h)


def encoder_forward(sd, dd, x, prefix="encoder."):
    """model.py:434-459; Downsample pads right/bottom only (:72-77)."""
    p = prefix
    nres = _levels(dd)
    curr_res = dd["resolution"]
    h = _conv(sd, p + "conv_in.", x)
    for lvl in range(nres):
        for ib in range(dd["num_res_blocks"]):
            h = resnet_block(sd, p + f"down.{lvl}.block.{ib}.", h)
            if curr_res in dd["attn_resolutions"]:
                h = attn_block(sd, p + f"down.{lvl}.attn.{ib}.", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(sd, p + f"down.{lvl}.downsample.conv.", h, stride=2, padding=0)
            curr_res //= 2
    h = resnet_block(sd, p + "mid.block_1.", h)
    h = attn_block(sd, p + "mid.attn_1.", h)
    h = resnet_block(sd, p + "mid.block_2.", h)
    h = _swish(_gn(sd, p + "norm_out.", h))
    return _conv(sd, p + "conv_out.", h)


def decode_first_stage(sd, dd, z, scale_factor=1.0):
    """ddpm_audio.py:352-359 (z/scale_factor) -> autoencoder.py:351-354."""
    z = z / scale_factor
    z = _conv(sd, "post_quant_conv.", z, padding=0)
    return decoder_forward(sd, dd, z)


def encode_moments(sd, dd, x):
    """autoencoder.py:345-349: Encoder -> quant_conv -> (mean, logvar clamp [-30, 20])."""
    h = encoder_forward(sd, dd, x)
    m = _conv(sd, "quant_conv.", h, padding=0)
    mean, logvar = torch.chunk(m, 2, dim=1)
    return mean, torch.clamp(logvar, -30.0, 20.0)


def posterior_sample(mean, logvar, noise):
    """distributions.py:35-37 with the randn supplied by the caller."""
    return mean + torch.exp(0.5 * logvar) * noise
