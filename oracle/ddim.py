"""Oracle: DDIM sampler + schedules, CPU fp32.  TEST INFRASTRUCTURE ONLY.

Restates (relative to /root/reference/text_to_audio/Make_An_Audio):
  ldm/modules/diffusionmodules/util.py:21-25   make_beta_schedule("linear") in fp64
  ldm/models/diffusion/ddpm.py:115-136         alphas_cumprod (fp64 cumprod -> fp32 buffer)
  ldm/modules/diffusionmodules/util.py:46-74   make_ddim_timesteps / make_ddim_sampling_parameters
  ldm/models/diffusion/ddim.py:27-56, 118-225  make_schedule, ddim_sampling, p_sample_ddim
"""
import numpy as np
import torch


def alphas_cumprod(timesteps=1000, linear_start=0.00085, linear_end=0.012):
    """fp32 tensor [timesteps], exactly as the registered buffer (ddpm.py:132-136)."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1.0 - betas, axis=0)
    return torch.tensor(ac, dtype=torch.float32)


def ddim_timesteps(S, T=1000):
    """util.py:46-59 ('uniform'): range(0, T, T//S) + 1."""
    c = T // S
    return np.asarray(list(range(0, T, c))) + 1


def ddim_tables(ac, steps, eta=0.0):
    """util.py:62-74 + ddim.py:46-52.  `ac` is the fp32 alphas_cumprod tensor.

    Returns fp32 tensors (alphas, alphas_prev, sigmas, sqrt_one_minus_alphas) indexed by DDIM index.
    """
    ac_c = ac.cpu()
    # the reference keeps alphas as an fp32 tensor and builds alphas_prev as a float64 numpy array
    # of python floats read from the fp32 buffer; sigmas are formed in that mixed arithmetic
    # (util.py:62-69): `ndarray / Tensor` resolves to Tensor.__rtruediv__ = self.reciprocal() * other, so (1 - alphas) AND its
    # reciprocal are rounded in fp32; everything else is fp64 (pinned by the eta = 0.5 golden, ddim_t2a_mask_eta_s6).
    a = ac_c[steps].numpy().astype(np.float32)
    ap = np.asarray([ac_c[0].item()] + ac_c[steps[:-1]].tolist())
    rec = (np.float32(1.0) / (np.float32(1.0) - a)).astype(np.float32)
    sig = eta * np.sqrt((1 - ap) * rec.astype(np.float64) * (1 - a.astype(np.float64) / ap))
    return (torch.tensor(a, dtype=torch.float32), torch.tensor(ap, dtype=torch.float32),
            torch.tensor(sig, dtype=torch.float32), torch.sqrt(1.0 - torch.tensor(a, dtype=torch.float32)))


def ddim_step(x, e_t, a_t, a_prev, sigma_t, sqrt_one_minus_at, noise=None):
    """ddim.py:210-225 with scalars broadcast.  Returns (x_prev, pred_x0)."""
    a_t = torch.full((x.shape[0], 1, 1, 1), float(a_t))
    a_prev = torch.full((x.shape[0], 1, 1, 1), float(a_prev))
    sigma_t = torch.full((x.shape[0], 1, 1, 1), float(sigma_t))
    somat = torch.full((x.shape[0], 1, 1, 1), float(sqrt_one_minus_at))
    pred_x0 = (x - somat * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    n = 0.0 if noise is None else sigma_t * noise
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + n
    return x_prev, pred_x0


def q_sample_tables(ac, steps):
    """ddpm.py:139-140: sqrt(alphas_cumprod) and sqrt(1 - alphas_cumprod) are formed in fp64 from the fp64 cumprod and stored as
    fp32 buffers; q_sample gathers them at the DDPM timestep (ddpm.py:272-275).  `ac` is the fp32 alphas_cumprod buffer (its
    values are the fp64 ones rounded once, which is what the samplers downstream of a checkpoint see as well)."""
    a = ac.cpu().numpy().astype(np.float64)[steps]
    return torch.tensor(np.sqrt(a), dtype=torch.float32), torch.tensor(np.sqrt(1.0 - a), dtype=torch.float32)


def ddim_sample(apply_model, ac, S, x_T, cond, uncond=None, scale=1.0, eta=0.0, noise_fn=None,
                trace=None, mask=None, x0=None, noise_q=None, noise_p=None, temperature=1.0, log_every_t=None,
                q_tables=None, score_fn=None, callback=None, img_callback=None):
    """ddim.py:118-166 + 169-225.

    apply_model(x, t, c) -> eps;  `cond`/`uncond` are tensors (crossattn context or concat cond).
    CFG batch order is [uncond ; cond] (ddim.py:177-199).
    mask / x0 / noise_q [S, ...]: img = q_sample(x0, t) * mask + (1 - mask) * img before every step (ddim.py:147-150), with the
    step's draw of randn_like(x0) given in loop order; noise_p [S, ...]: the steps' noise_like draws (ddim.py:221), scaled by
    sigma_t and `temperature`; log_every_t: also return the intermediates dict of ddim.py:138, 161-163.
    Host code inside the loop: score_fn(e_t, x, ts, cond) -> e_t stands for `score_corrector.modify_score(model, e_t, x, t, c,
    **corrector_kwargs)` (ddim.py:201-203, after the guidance mix); callback(i) and img_callback(pred_x0, i) run after every step
    (ddim.py:155-156), i counting from 0.
    """
    steps = ddim_timesteps(S, ac.shape[0])
    alphas, alphas_prev, sigmas, somas = ddim_tables(ac, steps, eta)
    sq_ac, sq_1mac = q_tables if q_tables is not None else q_sample_tables(ac, steps)
    x = x_T
    b = x.shape[0]
    total = steps.shape[0]
    inter = {"x_inter": [x], "pred_x0": [x]}
    for i, step in enumerate(np.flip(steps)):
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        if mask is not None:
            assert x0 is not None
            img_orig = sq_ac[index] * x0 + sq_1mac[index] * noise_q[i]
            x = img_orig * mask + (1.0 - mask) * x
        if uncond is None or scale == 1.0:
            e_t = apply_model(x, ts, cond)
        else:
            x_in = torch.cat([x] * 2)
            t_in = torch.cat([ts] * 2)
            c_in = torch.cat([uncond, cond])
            e_u, e_c = apply_model(x_in, t_in, c_in).chunk(2)
            e_t = e_u + scale * (e_c - e_u)
        if score_fn is not None:
            e_t = score_fn(e_t, x, ts, cond)
        if noise_p is not None:
            noise = noise_p[i] if eta > 0 else None
        else:
            noise = noise_fn(x.shape) if (noise_fn is not None and eta > 0) else None
        if noise is not None and temperature != 1.0:
            # ddim.py:221: sigma_t * noise * temperature, left to right
            sig = torch.full((b, 1, 1, 1), float(sigmas[index]))
            x_prev, pred = ddim_step(x, e_t, alphas[index], alphas_prev[index], 0.0, somas[index], None)
            a_prev = torch.full((b, 1, 1, 1), float(alphas_prev[index]))
            dir_xt = (1.0 - a_prev - sig ** 2).sqrt() * e_t
            x, x0_pred = a_prev.sqrt() * pred + dir_xt + sig * noise * temperature, pred
        else:
            x, x0_pred = ddim_step(x, e_t, alphas[index], alphas_prev[index], sigmas[index], somas[index], noise)
        if trace is not None:
            trace.append(x.clone())
        if callback:
            callback(i)
        if img_callback:
            img_callback(x0_pred, i)
        if log_every_t is not None and (index % log_every_t == 0 or index == total - 1):
            inter["x_inter"].append(x)
            inter["pred_x0"].append(x0_pred)
    if log_every_t is not None:
        return x, inter
    return x
