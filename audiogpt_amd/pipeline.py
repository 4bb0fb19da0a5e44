"""Make-An-Audio generation pipeline on one MI355X: DDIM (UNet) -> VAE decode -> clamp -> vocoder.

This is the body of the reference's `T2A.txt2audio` / `I2A.img2audio` / `Inpaint.inpaint`
(audio-chatgpt.py:158-183, 232-261, 500-528) between conditioning and waveform, run by
libaudiogpt_mi355x.  Batched over prompts: where the reference loops `vocoder.vocode(spec)` per sample
(audio-chatgpt.py:179-181) the vocoder here takes the whole batch (identical per-sample results).
"""
import numpy as np
import torch

from . import config as C
from . import weights as WT
from .backend import Context, UNet, VAE, Vocoder


def make_beta_schedule_linear(timesteps, linear_start, linear_end):
    """util.py:21-25 ("linear"): linspace(sqrt(b0), sqrt(b1), T)**2 in fp64."""
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2


def alphas_cumprod_f32(timesteps, linear_start, linear_end):
    """ddpm.py:115-136: fp64 cumprod stored as an fp32 buffer."""
    betas = make_beta_schedule_linear(timesteps, linear_start, linear_end)
    return np.cumprod(1.0 - betas, axis=0).astype(np.float32)


def ddim_schedule(S, ac_f32):
    """util.py:46-74 ('uniform', eta = 0): (timesteps, alphas, alphas_prev) as the sampler's tables."""
    T = ac_f32.shape[0]
    c = T // S
    steps = np.asarray(list(range(0, T, c))) + 1
    alphas = ac_f32[steps]
    alphas_prev = np.asarray([ac_f32[0]] + ac_f32[steps[:-1]].tolist(), dtype=np.float32)
    return steps, alphas, alphas_prev


class MakeAnAudio:
    """UNet + VAE + vocoder replicas on one device."""

    def __init__(self, device="cuda:0", ldm=None, vocoder_cfg=None, unet_sd=None, vae_sd=None, vocoder_sd=None,
                 seeds=(0, 1, 2), with_encoder=False, precision="f32", stream=None):
        # stream: a torch.cuda.Stream this replica runs on (its library context shares it).  None: the library creates its
        # own blocking stream, which orders against PyTorch's legacy default stream -- simple, but every default-stream op
        # is then a barrier across ALL replicas of the device.  Several replicas side by side want one stream each.
        self.ldm = ldm or C.LDM_T2A
        self.vocoder_cfg = vocoder_cfg or C.HIFIGAN_16K
        self.precision = precision
        self.stream = stream
        self.ctx = Context(device, stream=stream, precision=precision)
        self.device = self.ctx.device
        unet_sd = unet_sd if unet_sd is not None else WT.make_unet_state_dict(self.ldm["unet"], seed=seeds[0])
        vae_sd = vae_sd if vae_sd is not None else WT.make_vae_state_dict(self.ldm["vae"], seed=seeds[1],
                                                                          with_encoder=with_encoder)
        vocoder_sd = vocoder_sd if vocoder_sd is not None else WT.make_vocoder_state_dict(self.vocoder_cfg, seed=seeds[2])
        self.unet = UNet(self.ctx, self.ldm["unet"], unet_sd)
        self.vae = VAE(self.ctx, self.ldm["vae"], vae_sd)
        self.vocoder = Vocoder(self.ctx, self.vocoder_cfg, vocoder_sd)
        self.scale_factor = float(self.ldm.get("scale_factor", 1.0))
        self.alphas_cumprod = alphas_cumprod_f32(self.ldm["timesteps"], self.ldm["linear_start"], self.ldm["linear_end"])

    # ---- stages ----------------------------------------------------------------------------------
    def sample_latents(self, x_T, cond=None, uncond=None, scale=1.0, S=100, concat=None, use_graph=True):
        steps, a, ap = ddim_schedule(S, self.alphas_cumprod)
        return self.unet.ddim_sample(x_T, steps, a, ap, cond=cond, uncond=uncond, scale=scale, concat=concat,
                                     use_graph=use_graph)

    def decode(self, z):
        """decode_first_stage then the tools' clamp((x+1)/2, 0, 1) (audio-chatgpt.py:175-176) -> [B,80,T]."""
        return self.vae.decode_spec(z, self.scale_factor)      # (the clamp runs in the decoder's last pass: no torch arithmetic)

    def vocode(self, spec):
        return self.vocoder(spec)[:, 0]

    def generate_here(self, x_T, cond=None, uncond=None, scale=1.0, S=100, concat=None, use_graph=True):
        """generate() on the CURRENT torch stream, which must be this replica's stream when it has one (the caller orders
        inputs and outputs against other streams itself: bench.py's worker threads)."""
        z = self.sample_latents(x_T, cond, uncond, scale, S, concat, use_graph)
        spec = self.decode(z)
        return self.vocode(spec), spec, z

    def generate(self, x_T, cond=None, uncond=None, scale=1.0, S=100, concat=None, use_graph=True):
        """x_T [B,4,h,w] -> (wav [B, T*hop], spec [B,80,T], z [B,4,h,w]); all on the device.  With a private stream the
        work is ordered after the caller's current stream on entry and the caller's stream after it on return."""
        if self.stream is None:
            return self.generate_here(x_T, cond, uncond, scale, S, concat, use_graph)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            out = self.generate_here(x_T, cond, uncond, scale, S, concat, use_graph)
        cur.wait_stream(self.stream)
        for t in out:
            t.record_stream(cur)
        return out

    def audio_seconds(self, n_clips, frames):
        return n_clips * frames * self.vocoder.hop / float(self.vocoder_cfg["sampling_rate"])

    def close(self):
        for o in (self.unet, self.vae, self.vocoder, self.ctx):
            o.close()
