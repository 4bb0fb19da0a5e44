"""DiffSinger's shallow-diffusion sampler with the reference's Python surface, backed by the HIP library.

Inference subset of `GaussianDiffusion` (NeuralSeq/modules/diff/shallow_diffusion_tts.py:66-283) as the T2S tool drives
it (audio-chatgpt.py:298-339 -> NeuralSeq/inference/svs/ds_e2e.py): the FastSpeech2 front end (`self.fs2`) that turns
phonemes into the conditioning `decoder_inp` and the coarse mel is outside the accelerated path and stays with the
caller; what runs on the device is the part that costs the time -- `denoise_fn` (DiffNet, 20 gated dilated residual
layers) inside the PLMS loop (`pndm_speedup`), K_step / pndm_speedup evaluations per utterance.

    gd = GaussianDiffusion(C.DIFFSINGER_DS1000, device="cuda:0", state_dict=ckpt_denoise_fn_sd, spec_min=..., spec_max=...)
    mel = gd.infer(fs2_mel [B, T, 80], cond [B, 256, T])          # == ret['mel_out'] of forward(..., infer=True)
"""
import numpy as np
import torch

from . import config as C
from . import weights as WT
from .backend import Context, DiffNet, default_precision


def linear_beta_schedule(timesteps, max_beta):
    """shallow_diffusion_tts.py:43-49."""
    return np.linspace(1e-4, max_beta, timesteps)


class GaussianDiffusion(object):
    def __init__(self, cfg=None, device="cuda:0", state_dict=None, spec_min=None, spec_max=None, ctx=None, precision=None,
                 seed=7):
        self.cfg = dict(cfg or C.DIFFSINGER_DS1000)
        self.ctx = ctx or Context(device, precision=precision or default_precision())
        self.device = self.ctx.device
        self.mel_bins = self.cfg["in_dims"]
        self.num_timesteps = int(self.cfg["timesteps"])
        self.K_step = int(self.cfg["K_step"])
        betas = linear_beta_schedule(self.num_timesteps, self.cfg["max_beta"])
        ac = np.cumprod(1.0 - betas, axis=0)
        to32 = lambda a: torch.tensor(a, dtype=torch.float32, device=self.device)   # noqa: E731  (:82-96: fp32 buffers)
        self.betas, self.alphas_cumprod = to32(betas), to32(ac)
        self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod = to32(np.sqrt(ac)), to32(np.sqrt(1.0 - ac))
        m = self.mel_bins
        self.spec_min = torch.as_tensor(spec_min if spec_min is not None else [-6.0] * m, dtype=torch.float32, device=self.device)[None, None, :m]
        self.spec_max = torch.as_tensor(spec_max if spec_max is not None else [1.5] * m, dtype=torch.float32, device=self.device)[None, None, :m]
        sd = state_dict if state_dict is not None else WT.make_diffnet_state_dict(self.cfg, seed=seed)
        sd = WT.strip_prefix(sd, "denoise_fn.") or sd
        self.denoise_fn = DiffNet(self.ctx, self.cfg, sd)

    # ---- shallow_diffusion_tts.py:203-208, 279-283
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        ex = lambda a: a[t].reshape(-1, 1, 1, 1)    # noqa: E731
        return ex(self.sqrt_alphas_cumprod) * x_start + ex(self.sqrt_one_minus_alphas_cumprod) * noise

    def norm_spec(self, x):
        return (x - self.spec_min) / (self.spec_max - self.spec_min) * 2 - 1

    def denorm_spec(self, x):
        return (x + 1) / 2 * (self.spec_max - self.spec_min) + self.spec_min

    # ---- :262-269 with :166-201, on the device
    def sample_plms(self, x, cond, K_step=None, interval=None, use_graph=True):
        """x [B, 1, M, T] at step K_step - 1 -> x_0."""
        K = self.K_step if K_step is None else int(K_step)
        iv = int(self.cfg["pndm_speedup"]) if interval is None else int(interval)
        return self.denoise_fn.plms_sample(x, cond, self.alphas_cumprod.cpu().numpy(), K, iv, use_graph=use_graph)

    @torch.no_grad()
    def infer(self, fs2_mels, cond, noise=None, gaussian_start=False, mel2ph=None):
        """The infer branch of forward (:244-276) after the FastSpeech2 front end: fs2_mels [B, T, M] (ret['mel_out'] of
        fs2), cond [B, H, T] (ret['decoder_inp'].transpose(1, 2)) -> mel_out [B, T, M]; with mel2ph [B, T] (singing) the
        frames that belong to no phoneme are zeroed (:273-274)."""
        fs2_mels = fs2_mels.to(device=self.device, dtype=torch.float32)
        x0 = self.norm_spec(fs2_mels).transpose(1, 2)[:, None, :, :]
        t = torch.tensor([self.K_step - 1], device=self.device).long()
        x = self.q_sample(x0, t, noise)
        if gaussian_start:
            x = torch.randn((cond.shape[0], 1, self.mel_bins, cond.shape[2]), device=self.device)
        x = self.sample_plms(x, cond)
        out = self.denorm_spec(x[:, 0].transpose(1, 2))
        if mel2ph is not None:
            out = out * (torch.as_tensor(mel2ph).to(out.device) > 0).float()[:, :, None]
        return out
