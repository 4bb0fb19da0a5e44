"""DDIMSampler with the reference's Python call signature, backed by the HIP library.

Drop-in for `ldm.models.diffusion.ddim.DDIMSampler` (text_to_audio/Make_An_Audio/ldm/models/diffusion/
ddim.py:12-18, 59-115) as the tool classes use it (audio-chatgpt.py:166-174, 245-253, 513-517):

    sampler = DDIMSampler(model)
    samples, intermediates = sampler.sample(S=..., conditioning=c, batch_size=n, shape=[4, 10, 78], verbose=False,
                                            unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                            x_T=start_code)

The whole trajectory runs on the device inside `maa_ddim_sample`.  Arguments the tools never pass and the
device loop does not implement (eta > 0, mask/x0 blending, score correctors, quantisation, callbacks) raise
NotImplementedError rather than being ignored.  `intermediates` holds only the start and end points (the tools
discard it).
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
        self.ddim_sigmas = torch.zeros(len(steps)) if ddim_eta == 0.0 else None
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - self.ddim_alphas)
        if verbose:
            print(f"Selected timesteps for ddim sampler: {steps}")

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0,
               noise_dropout=0.0, score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1.0, unconditional_conditioning=None, **kwargs):
        if eta != 0.0 or mask is not None or x0 is not None or score_corrector is not None or quantize_x0 \
                or callback is not None or img_callback is not None or noise_dropout != 0.0:
            raise NotImplementedError("the device DDIM loop covers the tools' call pattern: eta=0, no mask/x0, "
                                      "no score corrector / quantisation / callbacks")
        if conditioning is not None and not isinstance(conditioning, dict):
            if conditioning.shape[0] != batch_size:
                print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        Cc, H, W = shape
        if x_T is None:
            # ddim.py:126-127 draws from the global torch RNG on the model's device
            x_T = torch.randn((batch_size, Cc, H, W), device=self.device)
        key = self.model.conditioning_key
        kw = dict(scale=float(unconditional_guidance_scale))
        if key == "concat":
            kw["concat"] = conditioning          # cat([x, c], dim=1) inside the loop (ddpm.py:1404-1406)
        else:
            kw["cond"] = conditioning
            kw["uncond"] = unconditional_conditioning
        img = self.model.unet.ddim_sample(x_T, self.ddim_timesteps, self.ddim_alphas.numpy(), self.ddim_alphas_prev,
                                          **kw)
        intermediates = {"x_inter": [x_T, img], "pred_x0": [x_T, img]}
        return img, intermediates
