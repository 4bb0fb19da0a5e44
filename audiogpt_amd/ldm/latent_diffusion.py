"""The model object the tools reach through `sampler.model`, with the reference's attribute surface.

Stands in for `ldm.models.diffusion.ddpm_audio.LatentDiffusion_audio` as used at inference
(audio-chatgpt.py:150-155, 161-175, 241-254, 501-518):
  .first_stage_model.embed_dim, .get_learned_conditioning, .cond_stage_model, .encode_first_stage,
  .get_first_stage_encoding, .decode_first_stage, .apply_model, .num_timesteps, .betas, .alphas_cumprod(_prev),
  .device, .load_state_dict(sd, strict=False), .to(device)
UNet and VAE run in libaudiogpt_mi355x; the conditioning encoders (CLAP text tower / OpenCLIP image tower,
ldm/modules/encoders/modules.py:173-212, 315-350) are outside the hot path: the caller
plugs one in as `cond_stage_model` (any object with `.encode(list[str])` / `.forward_img(img)`); without one a
deterministic synthetic embedder of the right shape is used so the plumbing runs end to end.
"""
import hashlib

import numpy as np
import torch

from .. import config as C
from .. import weights as WT
from ..backend import Context, UNet, VAE, default_precision
from ..pipeline import alphas_cumprod_f32, make_beta_schedule_linear


class DiagonalGaussianDistribution(object):
    """ldm/modules/distributions/distributions.py:24-62 over device moments."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape, device=self.parameters.device)

    def mode(self):
        return self.mean


class SyntheticEmbedder(object):
    """Deterministic stand-in for FrozenCLAPEmbedder / FrozenGlobalNormOpenCLIPEmbedder outputs:
    text -> layer-normed N(0,1) [B, 77, 1024] (CLAP Projection ends in a LayerNorm, CLAP/clap.py:19);
    image -> L2-normalised [B, 1, 1024] (modules.py:344-347).  Seeded by a hash of the input."""

    def __init__(self, tokens=77, dim=1024, device="cpu"):
        self.tokens, self.dim, self.device = tokens, dim, device

    def _seed(self, obj):
        return int.from_bytes(hashlib.sha256(repr(obj).encode()).digest()[:4], "little")

    def encode(self, texts):
        rows = []
        for t in texts:
            g = torch.Generator().manual_seed(self._seed(t))
            rows.append(torch.nn.functional.layer_norm(torch.randn(self.tokens, self.dim, generator=g), (self.dim,)))
        return torch.stack(rows).to(self.device)

    def __call__(self, texts):
        return self.encode(texts)

    def preprocess(self, image):
        return torch.from_numpy(np.asarray(image, dtype=np.float32).copy())

    def forward_img(self, image):
        g = torch.Generator().manual_seed(self._seed(tuple(image.shape)) ^ int(float(image.float().sum()) * 1000) % (2 ** 31))
        v = torch.randn(image.shape[0] if image.dim() == 4 else 1, 1, self.dim, generator=g)
        return (v / v.norm(dim=-1, keepdim=True)).to(self.device)

    def to(self, device):
        self.device = device
        return self


class _FirstStage(object):
    def __init__(self, owner, embed_dim):
        self._owner = owner
        self.embed_dim = embed_dim

    def decode(self, z):
        return self._owner.vae.decode(z, 1.0)

    def encode(self, x):
        return DiagonalGaussianDistribution(self._owner.vae.encode_moments(x))


class LatentDiffusionAudio(object):
    def __init__(self, ldm_config=None, device="cuda:0", state_dict=None, seeds=(0, 1), cond_stage_model=None,
                 precision=None, tokenizer=None, preprocess=None):
        self.cfg = ldm_config or C.LDM_T2A
        self.conditioning_key = self.cfg["conditioning_key"]
        self.num_timesteps = self.cfg["timesteps"]
        betas = make_beta_schedule_linear(self.num_timesteps, self.cfg["linear_start"], self.cfg["linear_end"])
        ac = alphas_cumprod_f32(self.num_timesteps, self.cfg["linear_start"], self.cfg["linear_end"])
        self.precision = precision or default_precision()
        self.ctx = Context(device, precision=self.precision)
        self.device = self.ctx.device
        self.betas = torch.tensor(betas, dtype=torch.float32, device=self.device)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32, device=self.device)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1].astype(np.float64)), dtype=torch.float32,
                                                device=self.device)
        # ddpm.py:139-140: formed from the fp64 cumprod, stored as fp32 buffers (q_sample and the sampler's mask blend read them)
        ac64 = np.cumprod(1.0 - betas, axis=0)
        self.sqrt_alphas_cumprod = torch.tensor(np.sqrt(ac64), dtype=torch.float32, device=self.device)
        self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac64), dtype=torch.float32, device=self.device)
        self.scale_factor = float(self.cfg.get("scale_factor", 1.0))
        self.unet = self.vae = None
        # conditioning encoder: the caller's; else the device towers when the checkpoint carries their weights
        # (`cond_stage_model.*`, ldm/encoders.py); else a seeded stand-in with the encoders' output statistics
        self._own_cond_stage = cond_stage_model is None
        self._tokenizer, self._preprocess = tokenizer, preprocess
        self.cond_stage_model = cond_stage_model or SyntheticEmbedder(
            tokens=77 if self.cfg["unet"]["variant"] == "t2a" else 1, device=self.device)
        self.first_stage_model = _FirstStage(self, self.cfg["vae"]["embed_dim"])
        if state_dict is None:
            state_dict = {}
            for k, v in WT.make_unet_state_dict(self.cfg["unet"], seed=seeds[0]).items():
                state_dict["model.diffusion_model." + k] = v
            for k, v in WT.make_vae_state_dict(self.cfg["vae"], seed=seeds[1]).items():
                state_dict["first_stage_model." + k] = v
        self.load_state_dict(state_dict, strict=False)

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:272-275: the forward process at (integer tensor) timestep t."""
        noise = torch.randn_like(x_start) if noise is None else noise
        shape = (t.shape[0],) + (1,) * (x_start.dim() - 1)
        return (self.sqrt_alphas_cumprod.gather(-1, t).reshape(shape) * x_start +
                self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(shape) * noise)

    # ---- nn.Module-like surface the tools touch ---------------------------------------------------
    def load_state_dict(self, sd, strict=False):
        """Accepts the reference checkpoint layout (ckpt["state_dict"], audio-chatgpt.py:150): UNet under
        `model.diffusion_model.`, VAE under `first_stage_model.`, optional `scale_factor` buffer
        (ddpm_audio.py:71-74).  Conditioning-encoder and schedule entries are ignored (strict=False)."""
        usd = WT.strip_prefix(sd, "model.diffusion_model.")
        vsd = WT.strip_prefix(sd, "first_stage_model.")
        if not usd or not vsd:
            raise KeyError("state_dict needs `model.diffusion_model.*` and `first_stage_model.*` entries")
        if self.unet is not None:
            self.unet.close()
            self.vae.close()
        self.unet = UNet(self.ctx, self.cfg["unet"], usd)
        self.vae = VAE(self.ctx, self.cfg["vae"], vsd)
        if "scale_factor" in sd:
            self.scale_factor = float(sd["scale_factor"])
        csd = WT.strip_prefix(sd, "cond_stage_model.")
        if csd and self._own_cond_stage and self.cfg["conditioning_key"] == "crossattn":
            from . import encoders as E          # (imports this module's Context users: keep it local)
            if any(k.startswith("caption_encoder.") for k in csd):          # FrozenCLAPEmbedder (T2A, txt2audio-cfg.yaml)
                self.cond_stage_model = E.FrozenCLAPEmbedder(state_dict=csd, tokenizer=self._tokenizer, ctx=self.ctx)
            elif any(k.startswith("model.visual.") for k in csd):           # FrozenGlobalNormOpenCLIPEmbedder (I2A)
                self.cond_stage_model = E.FrozenGlobalNormOpenCLIPEmbedder(state_dict=csd, preprocess=self._preprocess,
                                                                           ctx=self.ctx)
        return [], []

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise RuntimeError("the HIP backend has no CPU path")
        return self

    def eval(self):
        return self

    # ---- conditioning ---------------------------------------------------------------------------
    def get_learned_conditioning(self, c):
        """ddpm_audio.py:166-177: cond_stage_model.encode(c) -> [B, 77, 1024] (T2A) / [B, 1, 1024] (I2A)."""
        enc = self.cond_stage_model.encode(c) if hasattr(self.cond_stage_model, "encode") else self.cond_stage_model(c)
        return enc.to(device=self.device, dtype=torch.float32)

    # ---- first stage ----------------------------------------------------------------------------
    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x.to(self.device))

    def get_first_stage_encoding(self, encoder_posterior):
        """ddpm_audio.py:157-164: posterior.sample() (global torch RNG, as the reference) * scale_factor."""
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample()
        else:
            z = encoder_posterior
        return self.scale_factor * z

    def decode_first_stage(self, z):
        """ddpm_audio.py:352-359 -> autoencoder.py:351-354: (1/scale_factor) z -> post_quant_conv -> Decoder."""
        return self.vae.decode(z.to(self.device), self.scale_factor)

    # ---- diffusion model ------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond):
        """ddpm_audio.py:561-570,657 + DiffusionWrapper (ddpm.py:1400-1409)."""
        if isinstance(cond, dict):
            key = "c_concat" if self.conditioning_key == "concat" else "c_crossattn"
            cond = cond[key]
        if isinstance(cond, (list, tuple)):
            cond = torch.cat(list(cond), 1)
        if self.conditioning_key == "concat":
            return self.unet(torch.cat([x_noisy.to(self.device), cond.to(self.device)], dim=1), t)
        return self.unet(x_noisy, t, cond)
