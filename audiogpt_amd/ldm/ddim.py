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
and handed to the device loop.

Host code inside the loop -- `score_corrector` / `corrector_kwargs` (ddim.py:201-203), `callback(i)` / `img_callback(pred_x0, i)`
(ddim.py:155-156), `noise_dropout` (ddim.py:222-223) and `quantize_x0` (ddim.py:213-214) -- cannot live in a captured device
loop; a call that passes any of them takes `_host_loop` below: the reference's loop, one step per iteration, with the UNet
passes through `maa_unet_forward` (`model.apply_model`) and the update through `maa_ddim_update`, the host hooks called exactly
where the reference calls them.  None of the three tools passes them; the tool path stays the device loop.  A non-uniform
discretisation raises NotImplementedError.
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
        host_hooks = score_corrector is not None or quantize_x0 or callback is not None or img_callback is not None \
            or noise_dropout != 0.0
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
        if host_hooks:
            return self._host_loop(conditioning, x_T, callback=callback, img_callback=img_callback, quantize_denoised=quantize_x0,
                                   mask=mask, x0=x0, noise_dropout=noise_dropout, temperature=temperature,
                                   score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, log_every_t=log_every_t,
                                   unconditional_guidance_scale=unconditional_guidance_scale,
                                   unconditional_conditioning=unconditional_conditioning, noise_q=noise_q, noise_p=noise_p)
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

    def _host_loop(self, cond, x_T, callback, img_callback, quantize_denoised, mask, x0, noise_dropout, temperature,
                   score_corrector, corrector_kwargs, log_every_t, unconditional_guidance_scale, unconditional_conditioning,
                   noise_q, noise_p):
        """ddim_sampling + p_sample_ddim (ddim.py:118-166, 169-225) one step per iteration, for the calls that put host code
        inside the loop.  Device work per step: the UNet pass(es) (`apply_model` -> maa_unet_forward; the guided step as one
        batch [uncond ; cond], ddim.py:177-199) and the update (maa_ddim_update); the mask blend, the noise term and whatever
        the corrector does are torch arithmetic on the model's device, as in the reference."""
        dev, unet = self.device, self.model.unet
        img = x_T.to(dev)
        b = img.shape[0]
        total = len(self.ddim_timesteps)
        alphas, alphas_prev = self.ddim_alphas.numpy(), np.asarray(self.ddim_alphas_prev, dtype=np.float32)
        sigmas = np.asarray(self.ddim_sigmas, dtype=np.float32)
        somas = self.ddim_sqrt_one_minus_alphas.numpy()
        scale, uc = float(unconditional_guidance_scale), unconditional_conditioning
        intermediates = {"x_inter": [img], "pred_x0": [img]}
        for i, step in enumerate(np.flip(np.asarray(self.ddim_timesteps))):
            index = total - i - 1
            ts = torch.full((b,), int(step), device=dev, dtype=torch.long)
            if mask is not None:
                img_orig = self.model.q_sample(x0.to(dev), ts, noise=noise_q[i])
                img = img_orig * mask.to(dev) + (1.0 - mask.to(dev)) * img
            e_u = e_c = None
            if uc is None or scale == 1.0:
                e_u = self.model.apply_model(img, ts, cond)
            else:
                x_in, t_in = torch.cat([img] * 2), torch.cat([ts] * 2)
                if isinstance(cond, dict):
                    assert isinstance(uc, dict)
                    c_in = {k: ([torch.cat([uc[k][j], cond[k][j]]) for j in range(len(cond[k]))] if isinstance(cond[k], list)
                                else torch.cat([uc[k], cond[k]])) for k in cond}
                elif isinstance(cond, list):
                    assert isinstance(uc, list)
                    c_in = [torch.cat([uc[j], cond[j]]) for j in range(len(cond))]
                else:
                    c_in = torch.cat([uc.to(dev), cond.to(dev)])
                e_u, e_c = self.model.apply_model(x_in, t_in, c_in).chunk(2)
            if score_corrector is not None:
                assert getattr(self.model, "parameterization", "eps") == "eps"
                e_t = e_u if e_c is None else e_u + scale * (e_c - e_u)
                e_u, e_c = score_corrector.modify_score(self.model, e_t, img, ts, cond, **(corrector_kwargs or {})), None
            if quantize_denoised:
                # ddim.py:213-214 needs a VQ first stage; Make-An-Audio's is the KL autoencoder, which has no `quantize` (the
                # reference raises the same AttributeError from this line)
                quantize = self.model.first_stage_model.quantize
                e_t = e_u if e_c is None else e_u + scale * (e_c - e_u)
                a_t, a_prev, sig = float(alphas[index]), float(alphas_prev[index]), float(sigmas[index])
                pred_x0 = (img - float(somas[index]) * e_t) / float(np.sqrt(np.float32(a_t)))
                pred_x0, _, *_ = quantize(pred_x0)
                x_prev = float(np.sqrt(np.float32(a_prev))) * pred_x0 + float(np.sqrt(np.float32(1.0 - a_prev - sig * sig))) * e_t
            else:
                x_prev, pred_x0 = unet.ddim_update(img, e_u.contiguous(), None if e_c is None else e_c.contiguous(), scale,
                                                   alphas[index], alphas_prev[index], sigmas[index], somas[index])
            noise = float(sigmas[index]) * noise_p[i].to(dev) * temperature
            if noise_dropout > 0.0:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
            img = x_prev + noise
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates
