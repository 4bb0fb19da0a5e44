"""DDIMSampler with the reference's Python call signature, backed by the HIP library.

Drop-in for `ldm.models.diffusion.ddim.DDIMSampler` (text_to_audio/Make_An_Audio/ldm/models/diffusion/
ddim.py:12-18, 59-115) as the tool classes use it (audio-chatgpt.py:166-174, 245-253, 513-517):

    sampler = DDIMSampler(model)
    samples, intermediates = sampler.sample(S=..., conditioning=c, batch_size=n, shape=[4, 10, 78], verbose=False,
                                            unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                            x_T=start_code)

The whole trajectory runs on the device inside `maa_ddim_sample`, including the parts of the signature the tools never
use but the reference implements as tensor arithmetic: `mask` / `x0` blending (ddim.py:147-150), `eta > 0` and
`temperature` (ddim.py:210-225) and the `x_inter` / `pred_x0` logs every `log_every_t` steps (ddim.py:158-163).  The
reference draws the per-step noise from torch's global RNG inside its Python loop -- `randn_like(x0)` for q_sample when a
mask is given, then `noise_like(x.shape)` in p_sample_ddim, every step, whatever eta is; here the same draws are made
up front in the same order on the model's device (so a seeded call consumes the generator exactly as the reference does)
and handed to the device loop.  What needs host code inside the loop or a second model raises NotImplementedError rather
than being ignored: score correctors, quantize_x0, noise_dropout, callbacks, non-uniform discretisation.
"""
import numpy as np
import torch

from ..pipeline import ddim_schedule


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.device = model.device

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=True):
        """ddim.py:27-56: same tables, kept as numpy/torch fp32 attributes under the reference names."""
        if ddim_discretize != "uniform":
            raise NotImplementedError("only the 'uniform' DDIM discretisation is used by the tools")
        ac = self.model.alphas_cumprod.detach().cpu().numpy().astype(np.float32)
        steps, alphas, alphas_prev = ddim_schedule(ddim_num_steps, ac)
        self.ddim_timesteps = steps
        self.ddim_alphas = torch.from_numpy(alphas.copy())
        self.ddim_alphas_prev = alphas_prev
        # util.py:62-69: sigmas in the reference's mixed arithmetic -- alphas is an fp32 tensor there and alphas_prev an fp64
        # array of python floats; `ndarray / Tensor` is Tensor.reciprocal() * ndarray, so (1 - alphas) and its reciprocal are fp32,
        # the rest fp64; the result is an fp64 ndarray that torch.full rounds to fp32 per step
        a64, ap64 = alphas.astype(np.float64), np.asarray(alphas_prev, dtype=np.float64)
        rec = (np.float32(1.0) / (np.float32(1.0) - alphas.astype(np.float32))).astype(np.float64)
        self.ddim_sigmas = ddim_eta * np.sqrt((1 - ap64) * rec * (1 - a64 / ap64))
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - self.ddim_alphas)
        if verbose:
            print(f"Selected timesteps for ddim sampler: {steps}")

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0,
               noise_dropout=0.0, score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1.0, unconditional_conditioning=None, **kwargs):
        if score_corrector is not None or quantize_x0 or callback is not None or img_callback is not None \
                or noise_dropout != 0.0:
            raise NotImplementedError("the device DDIM loop has no host code inside it: no score corrector / quantisation / "
                                      "dropout noise / callbacks")
        if mask is not None and x0 is None:
            raise AssertionError("mask needs x0")          # ddim.py:148
        if conditioning is not None and not isinstance(conditioning, dict):
            if conditioning.shape[0] != batch_size:
                print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        Cc, H, W = shape
        size = (batch_size, Cc, H, W)
        if x_T is None:
            # ddim.py:126-127 draws from the global torch RNG on the model's device
            x_T = torch.randn(size, device=self.device)
        # the loop's draws, in its order (ddim.py:149 -> ddpm.py:273, then ddim.py:221 -> util.py:264-267), made up front.
        # `_step_noise=(noise_q or None, noise_p)` (a keyword of this port, for tests) supplies them instead.
        given = kwargs.pop("_step_noise", None)
        if given is not None:
            noise_q, noise_p = given
        else:
            nq, npp = [], []
            for _ in range(len(self.ddim_timesteps)):       # (range(0, T, T // S): S = 6 gives seven steps, as in the reference)
                if mask is not None:
                    nq.append(torch.randn(size, device=self.device))
                npp.append(torch.randn(size, device=self.device))
            noise_q = torch.stack(nq) if nq else None
            noise_p = torch.stack(npp)
        key = self.model.conditioning_key
        kw = dict(scale=float(unconditional_guidance_scale), log_every_t=int(log_every_t), temperature=float(temperature))
        if key == "concat":
            kw["concat"] = conditioning          # cat([x, c], dim=1) inside the loop (ddpm.py:1404-1406)
        else:
            kw["cond"] = conditioning
            kw["uncond"] = unconditional_conditioning
        if mask is not None:
            steps = np.asarray(self.ddim_timesteps)
            # q_sample's buffers at the DDPM timesteps of the loop (ddpm.py:139-140, 272-275)
            kw.update(mask=mask, x0=x0, noise_q=noise_q,
                      sqrt_ac=self.model.sqrt_alphas_cumprod.detach().cpu().numpy()[steps],
                      sqrt_1mac=self.model.sqrt_one_minus_alphas_cumprod.detach().cpu().numpy()[steps])
        if eta != 0.0:
            kw.update(sigmas=np.asarray(self.ddim_sigmas, dtype=np.float32), noise_p=noise_p)
        img, x_log, x0_log = self.model.unet.ddim_sample(x_T, self.ddim_timesteps, self.ddim_alphas.numpy(),
                                                         self.ddim_alphas_prev, **kw)
        # ddim.py:138, 161-163: the start point, then the logged steps
        intermediates = {"x_inter": [x_T] + list(x_log), "pred_x0": [x_T] + list(x0_log)}
        return img, intermediates
