"""Generate golden vectors from the REFERENCE's own modules (run in the build container only).

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.npz

The reference ships no tests or fixtures for this path, so the oracle is
pinned against the reference modules themselves: each case builds the reference module from
/root/reference, loads the seeded weights produced by `audiogpt_amd.weights` (reference key layout,
`strict=True`), runs it on seeded inputs on CPU fp32 and stores inputs + outputs.  Weights are NOT
stored (they are regenerated from the seed); `manifest.json` records every state_dict key/shape of
the reference modules so the factory's key layout is pinned too.

Import shims: `omegaconf` stub (bigvgan/models.py:17, openaimodel.py:474),
`scipy.signal.kaiser` shim and a `utils.hparams` stub for NeuralSeq's hifigan.py import chain.
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
MAA = os.path.join(REF, "text_to_audio", "Make_An_Audio")
NS = os.path.join(REF, "NeuralSeq")

from audiogpt_amd import config as C          # noqa: E402
from audiogpt_amd import weights as WT         # noqa: E402


def _install_shims():
    om = types.ModuleType("omegaconf")
    om.OmegaConf = type("OmegaConf", (), {"load": staticmethod(lambda p: None)})
    lc = types.ModuleType("omegaconf.listconfig")
    lc.ListConfig = type("ListConfig", (list,), {})
    om.listconfig = lc
    sys.modules.setdefault("omegaconf", om)
    sys.modules.setdefault("omegaconf.listconfig", lc)
    import scipy.signal
    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    sys.path.insert(0, MAA)


def _ns_generator_cls():
    """Load NeuralSeq/modules/hifigan/hifigan.py without its parallel_wavegan import chain."""
    import importlib.util
    for name in ("modules", "modules.parallel_wavegan", "modules.parallel_wavegan.layers",
                 "modules.parallel_wavegan.models", "modules.parallel_wavegan.models.source"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["modules.parallel_wavegan.layers"].UpsampleNetwork = object
    sys.modules["modules.parallel_wavegan.layers"].ConvInUpsampleNetwork = object
    sys.modules["modules.parallel_wavegan.models.source"].SourceModuleHnNSF = object
    spec = importlib.util.spec_from_file_location("ns_hifigan", os.path.join(NS, "modules/hifigan/hifigan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.HifiGanGenerator


def _manifest(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def _cond(n, L, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(n, L, 1024, generator=g)
    return torch.nn.functional.layer_norm(c, (1024,))


def unet_case(name, cfg, H, W, ctx_len, manifest, n=2, seed=0, save=True):
    if cfg["variant"] == "i2a":
        from ldm.modules.diffusionmodules.custom_openaimodel import UNetModel
        kw = dict(use_context_project=False)
    else:
        from ldm.modules.diffusionmodules.openaimodel import UNetModel
        kw = dict(legacy=cfg["legacy"])
    m = UNetModel(image_size=32, in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                  model_channels=cfg["model_channels"], attention_resolutions=list(cfg["attention_resolutions"]),
                  num_res_blocks=cfg["num_res_blocks"], channel_mult=list(cfg["channel_mult"]),
                  num_heads=cfg["num_heads"], num_head_channels=cfg["num_head_channels"],
                  use_spatial_transformer=cfg["use_spatial_transformer"],
                  transformer_depth=cfg["transformer_depth"], context_dim=cfg["context_dim"],
                  resblock_updown=cfg["resblock_updown"], use_checkpoint=True, **kw).eval()
    manifest[name] = _manifest(m)
    sd = WT.make_unet_state_dict(cfg, seed=seed)
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(n, cfg["in_channels"], H, W, generator=g)
    t = torch.tensor([981, 11][:n], dtype=torch.long)
    ctx = _cond(n, ctx_len, 1234) if cfg["context_dim"] else None
    with torch.no_grad():
        y = m(x, t, context=ctx)
    out = dict(x=x.numpy(), t=t.numpy(), y=y.numpy())
    if ctx is not None:
        out["context"] = ctx.numpy()
    if save:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "out std", float(y.std()), "absmax", float(y.abs().max()))
    return m


def ddim_case(name, unet, cfg, manifest, S=10, scale=1.5, seed=0):
    """DDIMSampler driven through a minimal model shim."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_beta_schedule

    class Shim:
        def __init__(self, ldm):
            betas = make_beta_schedule("linear", ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
            ac = np.cumprod(1.0 - betas, axis=0)
            self.num_timesteps = ldm["timesteps"]
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x, t, c):          # DiffusionWrapper crossattn path (ddpm.py:1407-1409)
            return unet(x, t, context=c)

    ldm = C.LDM_T2A
    sampler = DDIMSampler(Shim(ldm))
    sampler.device = torch.device("cpu")
    x_T = torch.from_numpy(np.random.RandomState(55).randn(1, 4, 10, 78)).float()   # audio-chatgpt.py:160-162
    c, uc = _cond(1, 77, 1234), _cond(1, 77, 1235)
    torch.manual_seed(0)
    with torch.no_grad():
        z, inter = sampler.sample(S=S, conditioning=c, batch_size=1, shape=[4, 10, 78], verbose=False,
                                  unconditional_guidance_scale=scale, unconditional_conditioning=uc, x_T=x_T,
                                  log_every_t=1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x_T=x_T.numpy(), c=c.numpy(), uc=uc.numpy(),
                        z=z.numpy(), ddim_timesteps=np.asarray(sampler.ddim_timesteps),
                        ddim_alphas=np.asarray(sampler.ddim_alphas), ddim_alphas_prev=np.asarray(sampler.ddim_alphas_prev),
                        x_inter=np.stack([t.numpy() for t in inter["x_inter"]]), S=S, scale=scale)
    print(name, "z std", float(z.std()))
    return z


def ddim_full_signature_case(name, unet, S=6, scale=1.5, eta=0.5, temperature=0.9, log_every_t=2, seed=31):
    """The rest of DDIMSampler.sample's signature on the T2A model: mask / x0 blending (ddim.py:147-150), eta > 0 with
    temperature (:210-225) and the intermediates every log_every_t steps (:158-163).  The sampler draws its per-step noise from
    torch's global RNG; the same draws are repeated afterwards from the same seed and stored (loop order), with the sampler's
    own sigma table and the q_sample buffers."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
    ldm = C.LDM_T2A

    class Shim:
        def __init__(self):
            betas = make_beta_schedule("linear", ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
            ac = np.cumprod(1.0 - betas, axis=0)
            self.num_timesteps = ldm["timesteps"]
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            # ddpm.py:139-140 (the DDPM class itself needs pytorch_lightning, which is not installed)
            self.sqrt_alphas_cumprod = torch.tensor(np.sqrt(ac), dtype=torch.float32)
            self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32)
            self.device = torch.device("cpu")

        def q_sample(self, x_start, t, noise=None):          # ddpm.py:272-275, verbatim semantics
            noise = torch.randn_like(x_start) if noise is None else noise
            return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                    extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

        def apply_model(self, x, t, c):
            return unet(x, t, context=c)

    shim = Shim()
    sampler = DDIMSampler(shim)
    sampler.device = torch.device("cpu")
    n = 2
    x_T = torch.from_numpy(np.random.RandomState(56).randn(n, 4, 10, 78)).float()
    g = torch.Generator().manual_seed(78)
    x0 = torch.randn(n, 4, 10, 78, generator=g)
    mask = torch.zeros(n, 1, 10, 78)
    mask[0, :, :, 20:45] = 1.0
    mask[1, :, 2:7, 50:] = 1.0
    c, uc = _cond(n, 77, 1234), _cond(n, 77, 1235)
    torch.manual_seed(seed)
    with torch.no_grad():
        z, inter = sampler.sample(S=S, conditioning=c, batch_size=n, shape=[4, 10, 78], verbose=False, eta=eta, mask=mask, x0=x0,
                                  temperature=temperature, unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                  x_T=x_T, log_every_t=log_every_t)
    torch.manual_seed(seed)
    nq, npp = [], []
    for _ in range(len(sampler.ddim_timesteps)):          # (S = 6 -> range(0, 1000, 166) = seven steps)
        nq.append(torch.randn_like(x0))
        npp.append(torch.randn(x_T.shape))
    steps = np.asarray(sampler.ddim_timesteps)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x_T=x_T.numpy(), x0=x0.numpy(), mask=mask.numpy(), c=c.numpy(), uc=uc.numpy(),
                        noise_q=torch.stack(nq).numpy(), noise_p=torch.stack(npp).numpy(), z=z.numpy(),
                        x_inter=np.stack([t.numpy() for t in inter["x_inter"]]),
                        pred_x0=np.stack([t.numpy() for t in inter["pred_x0"]]),
                        ddim_timesteps=steps, ddim_sigmas=np.asarray(sampler.ddim_sigmas, dtype=np.float64),
                        sqrt_ac=shim.sqrt_alphas_cumprod.numpy()[steps], sqrt_1mac=shim.sqrt_one_minus_alphas_cumprod.numpy()[steps],
                        S=S, scale=scale, eta=eta, temperature=temperature, log_every_t=log_every_t, seed=seed)
    print(name, "z std", float(z.std()), "logged", len(inter["x_inter"]))


def ddim_host_hooks_case(name, unet, S=4, scale=1.5, eta=0.3, seed=41):
    """The parts of DDIMSampler.sample that put HOST code inside the loop (ddim.py:154-156, 201-203): a score corrector
    (`modify_score(model, e_t, x, t, c, **corrector_kwargs)`), `callback(i)` and `img_callback(pred_x0, i)`, on the T2A model with
    guidance and eta > 0.  The corrector is an affine map of (e_t, x) so that the stored result pins the call order and the
    arguments; the per-step noise draws are repeated afterwards from the same seed, as in ddim_full_signature_case."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    ldm = C.LDM_T2A

    class Shim:
        parameterization = "eps"            # asserted by p_sample_ddim when a corrector is given (ddim.py:202)

        def __init__(self):
            betas = make_beta_schedule("linear", ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
            ac = np.cumprod(1.0 - betas, axis=0)
            self.num_timesteps = ldm["timesteps"]
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x, t, c):
            return unet(x, t, context=c)

    class Corrector:
        def modify_score(self, model, e_t, x, t, c, gain, shift):
            assert isinstance(model, Shim) and t.dtype == torch.long and c.shape[0] == x.shape[0]
            return gain * e_t + shift * x * (t.float() / 1000.0).reshape(-1, 1, 1, 1)

    shim = Shim()
    sampler = DDIMSampler(shim)
    sampler.device = torch.device("cpu")
    n = 2
    x_T = torch.from_numpy(np.random.RandomState(57).randn(n, 4, 10, 78)).float()
    c, uc = _cond(n, 77, 1234), _cond(n, 77, 1235)
    seen, preds = [], []
    torch.manual_seed(seed)
    with torch.no_grad():
        z, inter = sampler.sample(S=S, conditioning=c, batch_size=n, shape=[4, 10, 78], verbose=False, eta=eta,
                                  unconditional_guidance_scale=scale, unconditional_conditioning=uc, x_T=x_T, log_every_t=3,
                                  score_corrector=Corrector(), corrector_kwargs=dict(gain=0.9, shift=0.05),
                                  callback=seen.append, img_callback=lambda p, i: preds.append((i, p.clone())))
    torch.manual_seed(seed)
    npp = [torch.randn(x_T.shape) for _ in range(len(sampler.ddim_timesteps))]
    assert seen == list(range(len(sampler.ddim_timesteps))) and [i for i, _ in preds] == seen
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x_T=x_T.numpy(), c=c.numpy(), uc=uc.numpy(),
                        noise_p=torch.stack(npp).numpy(), z=z.numpy(), pred_x0_steps=np.stack([p.numpy() for _, p in preds]),
                        x_inter=np.stack([t.numpy() for t in inter["x_inter"]]), callback_i=np.asarray(seen),
                        S=S, scale=scale, eta=eta, gain=0.9, shift=0.05, log_every_t=3, seed=seed)
    print(name, "z std", float(z.std()), "steps", len(seen), "logged", len(inter["x_inter"]))


def ddim_variant_case(name, unet, ldm, S, scale, ctx_len=None):
    """The reference DDIMSampler on the other two tools' call patterns:
    inpaint -- conditioning_key 'concat' (ddpm.py:1404-1406: unet(cat([x] + [c], 1), t)), no guidance (audio-chatgpt.py:513-518);
    I2A     -- crossattn with a 1-token context and unconditional_guidance_scale 3 (audio-chatgpt.py:245-252)."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    concat = ldm["conditioning_key"] == "concat"

    class Shim:
        def __init__(self):
            betas = make_beta_schedule("linear", ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
            ac = np.cumprod(1.0 - betas, axis=0)
            self.num_timesteps = ldm["timesteps"]
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x, t, c):
            if concat:
                return unet(torch.cat([x] + [c], dim=1), t)
            return unet(x, t, context=c)

    sampler = DDIMSampler(Shim())
    sampler.device = torch.device("cpu")
    Cz, H, W = ldm["latent_shape"]
    x_T = torch.from_numpy(np.random.RandomState(55).randn(1, Cz, H, W)).float()
    out = dict(x_T=x_T.numpy(), S=S, scale=scale)
    if concat:
        g = torch.Generator().manual_seed(77)
        masked = torch.randn(1, Cz, H, W, generator=g)                     # encoded masked mel
        mask = torch.zeros(1, 1, H, W)
        mask[:, :, :, W // 3: W // 2] = 1.0                                # the region to fill (make_batch_sd)
        c = torch.cat([masked * (1 - mask), mask], dim=1)                  # audio-chatgpt.py:507-511
        out["c"] = c.numpy()
        kw = dict(conditioning=c)
    else:
        c, uc = _cond(1, ctx_len, 1234), _cond(1, ctx_len, 1235)           # image embedding / embedding of "" (audio-chatgpt.py:238-243)
        out["c"], out["uc"] = c.numpy(), uc.numpy()
        kw = dict(conditioning=c, unconditional_guidance_scale=scale, unconditional_conditioning=uc)
    with torch.no_grad():
        z, _ = sampler.sample(S=S, batch_size=1, shape=[Cz, H, W], verbose=False, x_T=x_T, **kw)
    out["z"] = z.numpy()
    out["ddim_timesteps"] = np.asarray(sampler.ddim_timesteps)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "z std", float(z.std()), "absmax", float(z.abs().max()))


def vae_case(name, dd, manifest, z=None, seed=1):
    from ldm.modules.diffusionmodules.model import Decoder, Encoder
    kw = dict(ch=dd["ch"], out_ch=dd["out_ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
              attn_resolutions=list(dd["attn_resolutions"]), in_channels=dd["in_channels"],
              resolution=dd["resolution"], z_channels=dd["z_channels"], double_z=dd["double_z"])
    dec, enc = Decoder(**kw).eval(), Encoder(**kw).eval()
    manifest[name + ".decoder"] = _manifest(dec)
    manifest[name + ".encoder"] = _manifest(enc)
    sd = WT.make_vae_state_dict(dd, seed=seed)
    dec.load_state_dict(WT.strip_prefix(sd, "decoder."), strict=True)
    enc.load_state_dict(WT.strip_prefix(sd, "encoder."), strict=True)
    pq = torch.nn.Conv2d(dd["embed_dim"], dd["z_channels"], 1)
    qc = torch.nn.Conv2d(2 * dd["z_channels"], 2 * dd["embed_dim"], 1)
    pq.load_state_dict(WT.strip_prefix(sd, "post_quant_conv."))
    qc.load_state_dict(WT.strip_prefix(sd, "quant_conv."))
    if z is None:
        g = torch.Generator().manual_seed(200 + seed)
        z = torch.randn(1, 4, 10, 78, generator=g)
    g = torch.Generator().manual_seed(300 + seed)
    mel_in = torch.rand(1, 1, 80, 848 if dd["resolution"] == 848 else 64, generator=g) * 2 - 1
    with torch.no_grad():
        mel = dec(pq(z))                  # autoencoder.py:351-354
        moments = qc(enc(mel_in))         # autoencoder.py:345-349
    np.savez_compressed(os.path.join(HERE, name + ".npz"), z=z.numpy(), mel=mel.numpy(),
                        mel_in=mel_in.numpy(), moments=moments.numpy())
    print(name, "mel std", float(mel.std()), "moments std", float(moments.std()))
    return mel


def hifigan_case(name, cfg, T, manifest, B=1, seed=2, mel=None):
    from argparse import Namespace
    from vocoder.hifigan.modules import Generator
    gen = Generator(Namespace(**{k: (list(map(list, v)) if k == "resblock_dilation_sizes" else
                                     (list(v) if isinstance(v, tuple) else v)) for k, v in cfg.items()})).eval()
    manifest[name + ".maa"] = _manifest(gen)
    sd = WT.make_vocoder_state_dict(cfg, seed=seed)
    gen.load_state_dict(sd, strict=True)
    if mel is None:
        g = torch.Generator().manual_seed(7)
        mel = torch.clamp(torch.randn(B, 80, T, generator=g) * 1.5 - 2.25, -6.0, 1.5)
    with torch.no_grad():
        wav = gen(mel)
    # the NeuralSeq generator must agree on the same weights (same graph, hifigan.py:144-169)
    NSGen = _ns_generator_cls()
    h = {k: (list(map(list, v)) if k == "resblock_dilation_sizes" else (list(v) if isinstance(v, tuple) else v))
         for k, v in cfg.items()}
    h["use_pitch_embed"] = False
    ns = NSGen(h).eval()
    manifest[name + ".ns"] = _manifest(ns)
    ns.load_state_dict(sd, strict=True)
    with torch.no_grad():
        wav_ns = ns(mel)
    assert torch.equal(wav, wav_ns), "MAA Generator and NeuralSeq HifiGanGenerator disagree"
    np.savez_compressed(os.path.join(HERE, name + ".npz"), mel=mel.numpy(), wav=wav.numpy())
    print(name, "wav std", float(wav.std()), "absmax", float(wav.abs().max()))
    return wav


def hifigan_nsf_case(name, cfg, T, manifest, B=2, seed=6):
    """NeuralSeq HifiGanGenerator with use_pitch_embed (the real SineGen / SourceModuleHnNSF of source.py), f0 given."""
    import importlib.util
    for n in ("modules", "modules.parallel_wavegan", "modules.parallel_wavegan.layers", "modules.parallel_wavegan.models"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["modules.parallel_wavegan.layers"].UpsampleNetwork = object
    sys.modules["modules.parallel_wavegan.layers"].ConvInUpsampleNetwork = object
    sp = importlib.util.spec_from_file_location("modules.parallel_wavegan.models.source",
                                                os.path.join(NS, "modules/parallel_wavegan/models/source.py"))
    src = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(src)
    sys.modules["modules.parallel_wavegan.models.source"] = src
    sp2 = importlib.util.spec_from_file_location("ns_hifigan_nsf", os.path.join(NS, "modules/hifigan/hifigan.py"))
    mod = importlib.util.module_from_spec(sp2)
    sp2.loader.exec_module(mod)
    h = {k: (list(map(list, v)) if k == "resblock_dilation_sizes" else (list(v) if isinstance(v, tuple) else v))
         for k, v in cfg.items()}
    h["audio_sample_rate"] = cfg["sampling_rate"]
    gen = mod.HifiGanGenerator(h).eval()
    manifest[name + ".ns"] = _manifest(gen)
    sd = WT.make_vocoder_state_dict(cfg, seed=seed)
    gen.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(9)
    mel = torch.clamp(torch.randn(B, 80, T, generator=g) * 1.5 - 2.25, -6.0, 1.5)
    # a voiced / unvoiced f0 contour: 110-440 Hz glides with unvoiced gaps
    t = torch.arange(T, dtype=torch.float32)
    f0 = torch.stack([220.0 * 2 ** (0.5 * torch.sin(2 * math.pi * t / (37.0 + 11 * b))) for b in range(B)])
    f0[:, T // 5: T // 4] = 0.0
    f0[1, -T // 6:] = 0.0
    torch.manual_seed(4321)                   # the oracle re-draws rand_ini / noise from this seed in the same order
    with torch.no_grad():
        wav = gen(mel, f0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), mel=mel.numpy(), f0=f0.numpy(), wav=wav.numpy(),
                        noise_seed=np.int64(4321))
    print(name, "wav std", float(wav.std()), "absmax", float(wav.abs().max()))
    return wav


def diffsinger_case(name, cfg, T, manifest, B=1, seed=7, K_step=60):
    """Reference DiffNet (modules/diff/net.py) + GaussianDiffusion.p_sample_plms driven exactly like the sampling loop of
    shallow_diffusion_tts.py:262-269 (gaussian start, pndm_speedup), on a shortened trajectory (K_step 60, interval 10).
    Batch 1: the reference's first PLMS step evaluates `max(t - interval, 0)` on the timestep tensor (:192), which only
    works for a single sample -- the T2S tool synthesises one utterance at a time."""
    import importlib.util
    from collections import deque
    hp = dict(hidden_size=cfg["hidden_size"], residual_layers=cfg["residual_layers"],
              residual_channels=cfg["residual_channels"], dilation_cycle_length=cfg["dilation_cycle_length"],
              max_beta=cfg["max_beta"], schedule_type="linear")
    stubs = {"utils": {}, "utils.hparams": {"hparams": hp}, "modules": {}, "modules.diff": {}, "modules.fastspeech": {},
             "modules.fastspeech.fs2": {"FastSpeech2": object}, "modules.diffsinger_midi": {},
             "modules.diffsinger_midi.fs2": {"FastSpeech2MIDI": object}}
    saved = {k: sys.modules.get(k) for k in stubs}
    for k, attrs in stubs.items():
        m = types.ModuleType(k)
        m.__path__ = []
        for a, v in attrs.items():
            setattr(m, a, v)
        sys.modules[k] = m

    def load(modname, rel):
        sp = importlib.util.spec_from_file_location(modname, os.path.join(NS, rel))
        m = importlib.util.module_from_spec(sp)
        sys.modules[modname] = m
        sp.loader.exec_module(m)
        return m
    try:
        load("modules.diff.diffusion", "modules/diff/diffusion.py")
        net = load("modules.diff.net", "modules/diff/net.py")
        sdt = load("modules.diff.shallow_diffusion_tts", "modules/diff/shallow_diffusion_tts.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    model = net.DiffNet(cfg["in_dims"]).eval()
    manifest[name] = _manifest(model)
    sd = WT.make_diffnet_state_dict(cfg, seed=seed)
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(21)
    cond = torch.randn(B, cfg["hidden_size"], T, generator=g)
    x_T = torch.randn(B, 1, cfg["in_dims"], T, generator=g)                 # gaussian_start (:257-260)
    t0 = torch.tensor([K_step - 10] * B, dtype=torch.long)
    with torch.no_grad():
        eps0 = model(x_T, t0, cond)

    betas = sdt.linear_beta_schedule(cfg["timesteps"], max_beta=cfg["max_beta"])
    ac = torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)

    class Shim:
        alphas_cumprod = ac
        noise_list = deque(maxlen=4)

        @staticmethod
        def denoise_fn(x, t, cond):
            return model(x, t, cond)
    x = x_T
    inter = []
    with torch.no_grad():
        for i in reversed(range(0, K_step, cfg["pndm_speedup"])):
            x = sdt.GaussianDiffusion.p_sample_plms(Shim, x, torch.full((B,), i, dtype=torch.long), cfg["pndm_speedup"], cond)
            inter.append(x.numpy().copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), cond=cond.numpy(), x_T=x_T.numpy(), t0=t0.numpy(), eps0=eps0.numpy(),
                        x0=x.numpy(), x_inter=np.stack(inter), K_step=K_step, alphas_cumprod=ac.numpy())
    print(name, "eps std", float(eps0.std()), "x0 std", float(x.std()), "absmax", float(x.abs().max()))


def bigvgan_case(name, cfg, T, manifest, seed=3):
    from argparse import Namespace
    from vocoder.bigvgan.models import BigVGAN
    gen = BigVGAN(Namespace(**{k: (list(map(list, v)) if k == "resblock_dilation_sizes" else
                                   (list(v) if isinstance(v, tuple) else v)) for k, v in cfg.items()})).eval()
    manifest[name] = _manifest(gen)
    sd = WT.make_vocoder_state_dict(cfg, seed=seed)
    missing, unexpected = gen.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("filter") for k in missing), missing      # registered FIR buffers only
    filt = gen.activation_post.upsample.filter.reshape(-1).numpy()
    g = torch.Generator().manual_seed(8)
    mel = torch.clamp(torch.randn(1, 80, T, generator=g) * 1.5 - 2.25, -6.0, 1.5)
    with torch.no_grad():
        wav = gen(mel)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), mel=mel.numpy(), wav=wav.numpy(), filter=filt)
    print(name, "wav std", float(wav.std()), "absmax", float(wav.abs().max()))


def config2_case(name, rows=(0, 5), S=100, scale=1.5, n_batch=8):
    """BASELINE configs[1] exactly as bench.py feeds it (8 prompts: x_T = RandomState(55).randn(8, 4, 10, 78), c =
    layer-normed N(0,1) rows of generator 1234, one unconditional row of generator 1235, CFG 1.5, 100 DDIM steps),
    for `rows` of that batch: the reference DDIMSampler over the reference UNetModel, then the reference VAE Decoder
    and HiFi-GAN Generator (audio-chatgpt.py:160-181 with the hifi_0127 vocoder).  ~3 minutes on 8 cores."""
    from argparse import Namespace
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.model import Decoder
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    from vocoder.hifigan.modules import Generator
    unet = unet_case("unet_t2a", C.UNET_T2A, 10, 78, 77, {}, save=False)

    class Shim:
        def __init__(self, ldm):
            betas = make_beta_schedule("linear", ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
            ac = np.cumprod(1.0 - betas, axis=0)
            self.num_timesteps = ldm["timesteps"]
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x, t, c):
            return unet(x, t, context=c)

    sampler = DDIMSampler(Shim(C.LDM_T2A))
    sampler.device = torch.device("cpu")
    rows = list(rows)
    x_T = torch.from_numpy(np.random.RandomState(55).randn(n_batch, 4, 10, 78)).float()[rows]
    c = _cond(n_batch, 77, 1234)[rows]
    uc = _cond(1, 77, 1235).expand(len(rows), -1, -1).contiguous()
    with torch.no_grad():
        z, _ = sampler.sample(S=S, conditioning=c, batch_size=len(rows), shape=[4, 10, 78], verbose=False,
                              unconditional_guidance_scale=scale, unconditional_conditioning=uc, x_T=x_T)
    dd = C.VAE_DDCONFIG
    dec = Decoder(ch=dd["ch"], out_ch=dd["out_ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
                  attn_resolutions=list(dd["attn_resolutions"]), in_channels=dd["in_channels"],
                  resolution=dd["resolution"], z_channels=dd["z_channels"], double_z=dd["double_z"]).eval()
    vsd = WT.make_vae_state_dict(dd, seed=1)
    dec.load_state_dict(WT.strip_prefix(vsd, "decoder."), strict=True)
    pq = torch.nn.Conv2d(dd["embed_dim"], dd["z_channels"], 1)
    pq.load_state_dict(WT.strip_prefix(vsd, "post_quant_conv."))
    cfg = C.HIFIGAN_16K
    gen = Generator(Namespace(**{k: (list(map(list, v)) if k == "resblock_dilation_sizes" else
                                     (list(v) if isinstance(v, tuple) else v)) for k, v in cfg.items()})).eval()
    gen.load_state_dict(WT.make_vocoder_state_dict(cfg, seed=2), strict=True)
    with torch.no_grad():
        spec = torch.clamp((dec(pq(z)) + 1.0) / 2.0, 0.0, 1.0)[:, 0]      # audio-chatgpt.py:175-176
        wav = gen(spec)[:, 0]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rows=np.asarray(rows), n_batch=n_batch, S=S, scale=scale,
                        z=z.numpy(), spec=spec.numpy().astype(np.float32), wav=wav.numpy().astype(np.float32))
    print(name, "z std", float(z.std()), "spec mean", float(spec.mean()), "wav std", float(wav.std()))


def config3_case(name, cfg, row, B=64, T=1024, seed=2):
    """BASELINE configs[2] input (mel [64, 80, 1024] = clip(N(-2.25, 1.5), -6, 1.5), generator seed 7, the formula of
    hifigan_case): ONE row of the batch through the reference NeuralSeq HifiGanGenerator (the full 64-row output is
    67 MB).  The mel is regenerated from the seed by the test; only the row's waveform is stored."""
    NSGen = _ns_generator_cls()
    h = {k: (list(map(list, v)) if k == "resblock_dilation_sizes" else (list(v) if isinstance(v, tuple) else v))
         for k, v in cfg.items()}
    h["use_pitch_embed"] = False
    ns = NSGen(h).eval()
    ns.load_state_dict(WT.make_vocoder_state_dict(cfg, seed=seed), strict=True)
    g = torch.Generator().manual_seed(7)
    mel = torch.clamp(torch.randn(B, 80, T, generator=g) * 1.5 - 2.25, -6.0, 1.5)
    with torch.no_grad():
        wav = ns(mel[row:row + 1])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), row=row, B=B, T=T, mel_seed=7, wav=wav.numpy().astype(np.float32))
    print(name, "wav std", float(wav.std()), "absmax", float(wav.abs().max()))


def _token_ids(B, L, vocab, seed):
    """[CLS] words ... [SEP] then [PAD] to L, like the tokenizer's padding="max_length" output (ids only matter as indices)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(3, L - 2, (1,), generator=g))
        ids[b, 0] = 101
        ids[b, 1:1 + n] = torch.randint(1000, vocab, (n,), generator=g)
        ids[b, 1 + n] = 102
    return ids


def clap_text_case(name, cfg, manifest, B=2, seed=11):
    """The reference's own chain (encoders/modules.py:208-211) on its own classes: transformers' BertModel (what
    AutoModel.from_pretrained('bert-base-uncased') instantiates, CLAP/clap.py:44; BertConfig() defaults are that model's
    dimensions) and the Projection class of the reference's CLAP/clap.py, loaded strict from the seeded factory."""
    import importlib.util
    from transformers import BertConfig, BertModel
    pkg = "refclap"
    stub = types.ModuleType(pkg)
    stub.__path__ = []
    audio = types.ModuleType(pkg + ".audio")
    audio.get_audio_encoder = lambda name: None
    sys.modules[pkg], sys.modules[pkg + ".audio"] = stub, audio
    sp = importlib.util.spec_from_file_location(pkg + ".clap", os.path.join(MAA, "ldm/modules/encoders/CLAP/clap.py"))
    clap = importlib.util.module_from_spec(sp)
    sys.modules[pkg + ".clap"] = clap
    sp.loader.exec_module(clap)
    bc = BertConfig()
    assert (bc.vocab_size, bc.hidden_size, bc.num_hidden_layers, bc.num_attention_heads, bc.intermediate_size,
            bc.max_position_embeddings, bc.type_vocab_size, bc.layer_norm_eps, bc.hidden_act) == \
        (cfg["vocab"], cfg["width"], cfg["layers"], cfg["heads"], cfg["mlp_dim"], cfg["max_positions"], cfg["type_vocab"],
         cfg["ln_eps"], "gelu")
    base = BertModel(bc, add_pooling_layer=False).eval()
    proj = clap.Projection(cfg["width"], cfg["d_proj"]).eval()
    sd = WT.make_clap_text_state_dict(cfg, seed=seed)
    missing = base.load_state_dict({k[len("base."):]: v for k, v in sd.items() if k.startswith("base.")}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("position_ids") for k in missing.missing_keys), missing
    proj.load_state_dict({k[len("projection."):]: v for k, v in sd.items() if k.startswith("projection.")}, strict=True)
    manifest[name] = {**{"base." + k: v for k, v in _manifest(base).items() if not k.endswith("position_ids")},
                      **{"projection." + k: v for k, v in _manifest(proj).items()}}
    ids = _token_ids(B, cfg["max_length"], cfg["vocab"], 31)
    with torch.no_grad():
        hidden = base(input_ids=ids).last_hidden_state          # modules.py:209
        z = proj(hidden)                                        # modules.py:210
    np.savez_compressed(os.path.join(HERE, name + ".npz"), input_ids=ids.numpy(), hidden_row0=hidden[0].numpy(), z=z.numpy())
    print(name, "hidden std", float(hidden.std()), "z std", float(z.std()), "absmax", float(z.abs().max()))


def openclip_image_case(name, cfg, manifest, B=2, seed=12):
    """open_clip is absent: the published ViT-H-14 image tower run through transformers' port of it
    (CLIPVisionModelWithProjection, hidden_act "gelu"), with the open_clip-layout seeded weights mapped onto its keys;
    then forward_img's normalisation (encoders/modules.py:341-343)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    W = cfg["width"]
    vc = CLIPVisionConfig(hidden_size=W, intermediate_size=cfg["mlp_dim"], projection_dim=cfg["d_proj"],
                          num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], image_size=cfg["image"],
                          patch_size=cfg["patch"], hidden_act="gelu", layer_norm_eps=cfg["ln_eps"], attention_dropout=0.0)
    try:
        vc._attn_implementation = "eager"
    except Exception:
        pass
    model = CLIPVisionModelWithProjection(vc).eval()
    sd = WT.make_openclip_visual_state_dict(cfg, seed=seed)
    hf = {"vision_model.embeddings.class_embedding": sd["class_embedding"],
          "vision_model.embeddings.patch_embedding.weight": sd["conv1.weight"],
          "vision_model.embeddings.position_embedding.weight": sd["positional_embedding"],
          "vision_model.pre_layrnorm.weight": sd["ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["ln_pre.bias"],
          "vision_model.post_layernorm.weight": sd["ln_post.weight"], "vision_model.post_layernorm.bias": sd["ln_post.bias"],
          "visual_projection.weight": sd["proj"].t().contiguous()}
    for i in range(cfg["layers"]):
        p, q = "transformer.resblocks.%d." % i, "vision_model.encoder.layers.%d." % i
        wi, bi = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            hf[q + "self_attn.%s.weight" % n] = wi[j * W:(j + 1) * W]
            hf[q + "self_attn.%s.bias" % n] = bi[j * W:(j + 1) * W]
        hf[q + "self_attn.out_proj.weight"], hf[q + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"]
        hf[q + "layer_norm1.weight"], hf[q + "layer_norm1.bias"] = sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]
        hf[q + "layer_norm2.weight"], hf[q + "layer_norm2.bias"] = sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]
        hf[q + "mlp.fc1.weight"], hf[q + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]
        hf[q + "mlp.fc2.weight"], hf[q + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"]
    missing = model.load_state_dict(hf, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("position_ids") for k in missing.missing_keys), missing
    manifest[name] = {k: list(v.shape) for k, v in sd.items()}
    g = torch.Generator().manual_seed(32)
    image = torch.randn(B, 3, cfg["image"], cfg["image"], generator=g)           # a preprocessed (normalised) image batch
    with torch.no_grad():
        z = model(pixel_values=image).image_embeds
        z = z / z.norm(dim=-1, keepdim=True)
        z = z.unsqueeze(1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), image_seed=32, z=z.numpy())
    print(name, "z", tuple(z.shape), "absmax", float(z.abs().max()))


def _clip_ids(cfg, B, seed):
    """Rows as open_clip.tokenize writes them: <start_of_text> tokens <end_of_text> 0 ... ; row 0 is the empty prompt."""
    g = torch.Generator().manual_seed(seed)
    L = cfg["max_positions"]
    ids = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = 0 if b == 0 else int(torch.randint(1, L - 2, (1,), generator=g))
        ids[b, 0] = cfg["sot"]
        ids[b, 1:1 + n] = torch.randint(300, cfg["sot"], (n,), generator=g)
        ids[b, 1 + n] = cfg["eot"]
    return ids


def openclip_text_case(name, cfg, manifest, B=3, seed=13):
    """open_clip is absent: CLIP.encode_text through transformers' port of it (CLIPTextModelWithProjection, hidden_act
    "gelu": causal mask, pooled at the end-of-text token) carrying the open_clip-layout seeded weights; then forward's
    normalisation (encoders/modules.py:337-338)."""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    W = cfg["width"]
    tc = CLIPTextConfig(vocab_size=cfg["vocab"], hidden_size=W, intermediate_size=cfg["mlp_dim"], projection_dim=cfg["d_proj"],
                        num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                        max_position_embeddings=cfg["max_positions"], hidden_act="gelu", layer_norm_eps=cfg["ln_eps"],
                        attention_dropout=0.0, bos_token_id=cfg["sot"], eos_token_id=cfg["eot"], pad_token_id=0)
    model = CLIPTextModelWithProjection(tc).eval()
    sd = WT.make_openclip_text_state_dict(cfg, seed=seed)
    hf = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
          "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
          "text_model.final_layer_norm.weight": sd["ln_final.weight"], "text_model.final_layer_norm.bias": sd["ln_final.bias"],
          "text_projection.weight": sd["text_projection"].t().contiguous()}
    for i in range(cfg["layers"]):
        p, q = "transformer.resblocks.%d." % i, "text_model.encoder.layers.%d." % i
        wi, bi = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            hf[q + "self_attn.%s.weight" % n] = wi[j * W:(j + 1) * W]
            hf[q + "self_attn.%s.bias" % n] = bi[j * W:(j + 1) * W]
        hf[q + "self_attn.out_proj.weight"], hf[q + "self_attn.out_proj.bias"] = sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"]
        hf[q + "layer_norm1.weight"], hf[q + "layer_norm1.bias"] = sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]
        hf[q + "layer_norm2.weight"], hf[q + "layer_norm2.bias"] = sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]
        hf[q + "mlp.fc1.weight"], hf[q + "mlp.fc1.bias"] = sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]
        hf[q + "mlp.fc2.weight"], hf[q + "mlp.fc2.bias"] = sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"]
    missing = model.load_state_dict(hf, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("position_ids") for k in missing.missing_keys), missing
    manifest[name] = {k: list(v.shape) for k, v in sd.items()}
    ids = _clip_ids(cfg, B, 33)
    with torch.no_grad():
        z = model(input_ids=ids).text_embeds
        z = z / z.norm(dim=-1, keepdim=True)
        z = z.unsqueeze(1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), input_ids=ids.numpy(), z=z.numpy())
    print(name, "z", tuple(z.shape), "absmax", float(z.abs().max()))


def main_clip_text_only():
    """`python tests/golden/make_golden.py cliptext`: the OpenCLIP text tower case only."""
    torch.set_num_threads(8)
    _install_shims()
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    openclip_text_case("openclip_vith14_text", C.OPENCLIP_VITH14_TEXT, manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("torch", torch.__version__)


def clap_audio_case(name, cfg, manifest, B=2, seed=14):
    """The reference's own Cnn14 / Projection classes (CLAP/audio.py, CLAP/clap.py) from the log-mel on: torchlibrosa is
    absent, so its two extractor classes are stubbed while the module is built and the forward is entered after them
    (audio.py:150-176 replayed on the reference module's own submodules)."""
    import importlib.util
    tl = types.ModuleType("torchlibrosa")
    st = types.ModuleType("torchlibrosa.stft")
    st.Spectrogram = st.LogmelFilterBank = lambda *a, **k: torch.nn.Identity()
    tl.stft = st
    sys.modules.setdefault("torchlibrosa", tl)
    sys.modules.setdefault("torchlibrosa.stft", st)
    pkg = "refclap2"
    stub = types.ModuleType(pkg)
    stub.__path__ = []
    sys.modules[pkg] = stub
    mods = {}
    for m in ("audio", "clap"):
        sp = importlib.util.spec_from_file_location(pkg + "." + m, os.path.join(MAA, "ldm/modules/encoders/CLAP/%s.py" % m))
        mods[m] = importlib.util.module_from_spec(sp)
        sys.modules[pkg + "." + m] = mods[m]
        sp.loader.exec_module(mods[m])
    enc = mods["clap"].AudioEncoder("Cnn14", cfg["out_emb"], cfg["d_proj"], cfg["sample_rate"], cfg["window_size"],
                                    cfg["hop_size"], cfg["mel_bins"], cfg["fmin"], cfg["fmax"], cfg["classes_num"]).eval()
    sd = WT.make_clap_audio_state_dict(cfg, seed=seed)
    enc.load_state_dict(sd, strict=True)
    manifest[name] = _manifest(enc)
    g = torch.Generator().manual_seed(34)
    logmel = torch.randn(B, 1, cfg["frames"], cfg["mel_bins"], generator=g) * 12.0 - 30.0      # dB-like values
    base = enc.base
    with torch.no_grad():
        x = base.bn0(logmel.transpose(1, 3)).transpose(1, 3)
        for i, blk in enumerate((base.conv_block1, base.conv_block2, base.conv_block3, base.conv_block4, base.conv_block5,
                                 base.conv_block6)):
            x = blk(x, pool_size=(2, 2) if i < 5 else (1, 1), pool_type="avg")
        x = torch.mean(x, dim=3)
        x = torch.max(x, dim=2)[0] + torch.mean(x, dim=2)
        emb = torch.relu(base.fc1(x))
        z = enc.projection(emb)
        z = z / torch.norm(z, dim=-1, keepdim=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), logmel_seed=34, embedding=emb.numpy(), z=z.numpy())
    print(name, "embedding", tuple(emb.shape), "absmax", float(emb.abs().max()), "z absmax", float(z.abs().max()))


def clap_score_case(name, cfg, n_audio=3, seed=21):
    """The best-of-n scorer on the reference's OWN wav_evaluation classes (T2A.select_best_audio, audio-chatgpt.py:185-199):
    TextEncoder (wav_evaluation/models/clap.py:41-53) fed what CLAPWrapper.preprocess_text builds -- input_ids padded to
    text_len, token_type_ids, attention_mask -- with `AutoModel.from_pretrained` answered by transformers' BertModel
    (what it instantiates for bert-base-uncased; no download here), AudioEncoder (clap.py:22-39 over
    wav_evaluation/models/audio.py's Cnn14) entered after its two torchlibrosa extractors (absent), the two
    normalisations of CLAPWrapper (:177-189) and compute_similarity(use_logit_scale=False) (:207-215).
    The clips are 9 * 16000 samples long at hop 320: 451 frames -- the scorer's own shape."""
    import importlib.util
    import transformers
    from transformers import BertConfig, BertModel
    tl = types.ModuleType("torchlibrosa")
    st = types.ModuleType("torchlibrosa.stft")
    st.Spectrogram = st.LogmelFilterBank = lambda *a, **k: torch.nn.Identity()
    tl.stft = st
    sys.modules.setdefault("torchlibrosa", tl)
    sys.modules.setdefault("torchlibrosa.stft", st)
    pkg = "refwaveval"
    stub = types.ModuleType(pkg)
    stub.__path__ = []
    sys.modules[pkg] = stub
    real_from_pretrained = transformers.AutoModel.from_pretrained
    transformers.AutoModel.from_pretrained = staticmethod(lambda name, *a, **k: BertModel(BertConfig()))
    try:
        mods = {}
        for m in ("audio", "clap"):
            sp = importlib.util.spec_from_file_location(pkg + "." + m, os.path.join(MAA, "wav_evaluation/models/%s.py" % m))
            mods[m] = importlib.util.module_from_spec(sp)
            sys.modules[pkg + "." + m] = mods[m]
            sp.loader.exec_module(mods[m])
        a = cfg["audio"]
        clap = mods["clap"].CLAP(audioenc_name="Cnn14", sample_rate=cfg["sampling_rate"], window_size=cfg["window_size"],
                                 hop_size=cfg["hop_size"], mel_bins=cfg["mel_bins"], fmin=cfg["fmin"], fmax=cfg["fmax"],
                                 classes_num=a["classes_num"], out_emb=a["out_emb"], text_model="bert-base-uncased",
                                 transformer_embed_dim=cfg["text"]["width"], d_proj=a["d_proj"]).eval()
    finally:
        transformers.AutoModel.from_pretrained = real_from_pretrained
    tsd = WT.make_clap_text_state_dict(cfg["text"], seed=seed)
    asd = WT.make_clap_audio_state_dict(a, seed=seed + 1)
    sd = {"caption_encoder." + k: v for k, v in tsd.items()}
    sd.update({"audio_encoder." + k: v for k, v in asd.items()})
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07))
    missing = clap.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("pooler" in k or k.endswith("position_ids") for k in missing.missing_keys), missing
    # preprocess_text's tensors for one prompt of 9 word pieces: [CLS] w1..w9 [SEP] [PAD]...
    L = cfg["text_len"]
    g = torch.Generator().manual_seed(41)
    n_tok = 9
    ids = torch.zeros(1, L, dtype=torch.long)
    ids[0, 0] = 101
    ids[0, 1:1 + n_tok] = torch.randint(1000, cfg["text"]["vocab"], (n_tok,), generator=g)
    ids[0, 1 + n_tok] = 102
    mask = torch.zeros(1, L, dtype=torch.long)
    mask[0, :n_tok + 2] = 1
    tok = {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}
    frames = 9 * 16000 // cfg["hop_size"] + 1
    # three clearly different "clips": level, spread and a spectral tilt per clip (dB-like values)
    scales = torch.tensor([12.0, 6.0, 18.0])[:n_audio]
    offsets = torch.tensor([-30.0, -10.0, -45.0])[:n_audio]
    tilts = torch.tensor([0.0, -0.4, 0.3])[:n_audio]
    logmel = torch.randn(n_audio, 1, frames, cfg["mel_bins"], generator=torch.Generator().manual_seed(42)) * \
        scales.view(-1, 1, 1, 1) + offsets.view(-1, 1, 1, 1) + tilts.view(-1, 1, 1, 1) * torch.arange(cfg["mel_bins"]).view(1, 1, 1, -1)
    with torch.no_grad():
        te = clap.caption_encoder(tok)                                                 # CLAPWrapper._get_text_embeddings
        te = te / torch.norm(te, dim=-1, keepdim=True)
        te = te / torch.norm(te, dim=-1, keepdim=True)                                 # (:181) a second time
        base = clap.audio_encoder.base
        x = base.bn0(logmel.transpose(1, 3)).transpose(1, 3)
        for i, blk in enumerate((base.conv_block1, base.conv_block2, base.conv_block3, base.conv_block4, base.conv_block5,
                                 base.conv_block6)):
            x = blk(x, pool_size=(2, 2) if i < 5 else (1, 1), pool_type="avg")
        x = torch.mean(x, dim=3)
        x = torch.max(x, dim=2)[0] + torch.mean(x, dim=2)
        emb = torch.relu(base.fc1(x))
        ae = clap.audio_encoder.projection(emb)
        ae = ae / torch.norm(ae, dim=-1, keepdim=True)
        ae = ae / torch.norm(ae, dim=-1, keepdim=True)
        sim = (te @ ae.T).T                                                            # compute_similarity, no logit scale
    np.savez_compressed(os.path.join(HERE, name + ".npz"), input_ids=ids[0, :n_tok + 2].numpy(), text_len=L, logmel_seed=42,
                        scales=scales.numpy(), offsets=offsets.numpy(), tilts=tilts.numpy(), frames=frames, text_embedding=te.numpy(), audio_embedding=ae.numpy(), similarity=sim.numpy())
    print(name, "similarity", sim.reshape(-1).tolist())


def main_clap_score_only():
    """`python tests/golden/make_golden.py clapscore`: the best-of-n scorer through the reference's wav_evaluation classes."""
    torch.set_num_threads(8)
    _install_shims()
    clap_score_case("clap_score", C.CLAP_SCORER)
    print("torch", torch.__version__)


def mixed_case(name, row_i2a=3, row_inp=6, S=100, n_batch=8):
    """BASELINE configs[4] as bench.py feeds it on one GPU (bench.mixed_inputs): ONE row of each tool through the reference's
    own classes, 100 DDIM steps.
      image-to-audio (audio-chatgpt.py:232-261): DDIMSampler over custom_openaimodel.UNetModel with a one-token context,
        guidance 3 -> Decoder -> clamp -> BigVGAN (624 frames)
      inpaint (:500-528): Encoder + quant_conv -> posterior sample (the row's noise) -> cat with the resized mask ->
        concat-conditioned DDIMSampler (x_T given) -> Decoder -> compositing with the input mel -> BigVGAN (848 frames)
    ~3 minutes on 8 cores."""
    from argparse import Namespace
    import bench
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.model import Decoder, Encoder
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    from vocoder.bigvgan.models import BigVGAN
    mel, mask, emb, uc, noise, xT_inp, xT_i2a = bench.mixed_inputs(n_batch)
    dd = C.VAE_DDCONFIG
    kw = dict(ch=dd["ch"], out_ch=dd["out_ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
              attn_resolutions=list(dd["attn_resolutions"]), in_channels=dd["in_channels"], resolution=dd["resolution"],
              z_channels=dd["z_channels"], double_z=dd["double_z"])
    dec, enc = Decoder(**kw).eval(), Encoder(**kw).eval()
    vsd = WT.make_vae_state_dict(dd, seed=1)
    dec.load_state_dict(WT.strip_prefix(vsd, "decoder."), strict=True)
    enc.load_state_dict(WT.strip_prefix(vsd, "encoder."), strict=True)
    pq = torch.nn.Conv2d(dd["embed_dim"], dd["z_channels"], 1)
    qc = torch.nn.Conv2d(2 * dd["z_channels"], 2 * dd["embed_dim"], 1)
    pq.load_state_dict(WT.strip_prefix(vsd, "post_quant_conv."))
    qc.load_state_dict(WT.strip_prefix(vsd, "quant_conv."))
    cfg = C.BIGVGAN_16K
    gen = BigVGAN(Namespace(**{k: (list(map(list, v)) if k == "resblock_dilation_sizes" else
                                   (list(v) if isinstance(v, tuple) else v)) for k, v in cfg.items()})).eval()
    missing, unexpected = gen.load_state_dict(WT.make_vocoder_state_dict(cfg, seed=3), strict=False)
    assert not unexpected and all(k.endswith("filter") for k in missing), (missing, unexpected)

    def sampler_for(unet, ldm):
        concat = ldm["conditioning_key"] == "concat"

        class Shim:
            def __init__(self):
                betas = make_beta_schedule("linear", ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
                ac = np.cumprod(1.0 - betas, axis=0)
                self.num_timesteps = ldm["timesteps"]
                self.betas = torch.tensor(betas, dtype=torch.float32)
                self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
                self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
                self.device = torch.device("cpu")

            def apply_model(self, x, t, c):
                return unet(torch.cat([x] + [c], dim=1), t) if concat else unet(x, t, context=c)
        sm = DDIMSampler(Shim())
        sm.device = torch.device("cpu")
        return sm

    out = dict(S=S, n_batch=n_batch, row_i2a=row_i2a, row_inp=row_inp)
    with torch.no_grad():
        r = row_i2a
        u = unet_case("unet_i2a", C.UNET_I2A, 10, 78, 1, {}, seed=4, save=False)
        z, _ = sampler_for(u, C.LDM_I2A).sample(S=S, conditioning=emb[r:r + 1], batch_size=1, shape=[4, 10, 78], verbose=False,
                                                unconditional_guidance_scale=3.0, unconditional_conditioning=uc[:1],
                                                x_T=xT_i2a[r:r + 1])
        spec = torch.clamp((dec(pq(z)) + 1.0) / 2.0, 0.0, 1.0)[:, 0]
        out.update(i2a_z=z.numpy(), i2a_spec=spec.numpy(), i2a_wav=gen(spec)[:, 0].numpy())
        r = row_inp
        u = unet_case("unet_inpaint", C.UNET_INPAINT, 10, 106, 0, {}, n=1, seed=5, save=False)
        m, k = mel[r:r + 1], mask[r:r + 1]
        mean, logvar = qc(enc((1 - k) * m * 2 - 1)).chunk(2, dim=1)                      # autoencoder.py:345-349
        zc = mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise[r:r + 1]  # distributions.py:34-47
        cc = torch.nn.functional.interpolate(k * 2 - 1, size=zc.shape[-2:])             # audio-chatgpt.py:510-511
        c = torch.cat((zc, cc), dim=1)
        z, _ = sampler_for(u, C.LDM_INPAINT).sample(S=S, conditioning=c, batch_size=1, shape=(4, 10, 106), verbose=False,
                                                    x_T=xT_inp[r:r + 1])
        pred = torch.clamp((dec(pq(z)) + 1.0) / 2.0, 0.0, 1.0)
        comp = ((1 - k) * m + k * pred)[:, 0]                                            # (:523-526)
        out.update(inp_z=z.numpy(), inp_spec=comp.numpy(), inp_wav=gen(comp)[:, 0].numpy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: (v.astype(np.float32) if isinstance(v, np.ndarray) else v)
                                                              for k, v in out.items()})
    print(name, "i2a z std", float(out["i2a_z"].std()), "wav std", float(out["i2a_wav"].std()),
          "| inpaint z std", float(out["inp_z"].std()), "wav std", float(out["inp_wav"].std()))


def main_mixed_only():
    """`python tests/golden/make_golden.py mixed`"""
    torch.set_num_threads(8)
    _install_shims()
    mixed_case("mixed_config5_s100")
    print("torch", torch.__version__)


def main_clap_audio_only():
    """`python tests/golden/make_golden.py clapaudio`: the CLAP audio-branch case (groundwork, SURVEY 8f / N4 scorer)."""
    torch.set_num_threads(8)
    _install_shims()
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    clap_audio_case("clap_audio_cnn14", C.CLAP_AUDIO_CNN14, manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("torch", torch.__version__)


def main_encoders_only():
    """`python tests/golden/make_golden.py encoders`: the conditioning-encoder cases."""
    torch.set_num_threads(8)
    _install_shims()
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    clap_text_case("clap_text_bert", C.CLAP_TEXT, manifest)
    openclip_image_case("openclip_vith14_image", C.OPENCLIP_VITH14_IMAGE, manifest)
    openclip_text_case("openclip_vith14_text", C.OPENCLIP_VITH14_TEXT, manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("torch", torch.__version__)


def main_config3_only():
    """`python tests/golden/make_golden.py config3`"""
    torch.set_num_threads(8)
    _install_shims()
    config3_case("hifigan_ns512_cfg3_row63", C.HIFIGAN_NS_512, 63)
    config3_case("hifigan_ns128_cfg3_row0", C.HIFIGAN_NS_128, 0)
    print("torch", torch.__version__)


def main_config2_only():
    """`python tests/golden/make_golden.py config2`: the benchmark configuration's own golden + the 624-frame BigVGAN."""
    torch.set_num_threads(8)
    _install_shims()
    config2_case("t2a_config2_s100")
    bigvgan_case("bigvgan_16k_t624", C.BIGVGAN_16K, 624, {})
    print("torch", torch.__version__)


def main():
    torch.set_num_threads(8)
    _install_shims()
    manifest = {}
    unet = unet_case("unet_t2a", C.UNET_T2A, 10, 78, 77, manifest)
    z = ddim_case("ddim_t2a_s10", unet, C.UNET_T2A, manifest)
    u_i2a = unet_case("unet_i2a", C.UNET_I2A, 10, 78, 1, manifest, seed=4)
    u_inp = unet_case("unet_inpaint", C.UNET_INPAINT, 10, 106, 0, manifest, n=1, seed=5)
    ddim_variant_case("ddim_i2a_s4", u_i2a, C.LDM_I2A, 4, 3.0, ctx_len=1)
    ddim_variant_case("ddim_inpaint_s4", u_inp, C.LDM_INPAINT, 4, 1.0)
    ddim_full_signature_case("ddim_t2a_mask_eta_s6", unet)
    ddim_host_hooks_case("ddim_t2a_host_hooks_s4", unet)
    mel = vae_case("vae", C.VAE_DDCONFIG, manifest, z=z)
    # plumbing config 1 end to end: clamp((x+1)/2, 0, 1) -> vocoder (audio-chatgpt.py:176-181)
    spec = torch.clamp((mel + 1.0) / 2.0, 0.0, 1.0)[:, 0]
    hifigan_case("hifigan_16k_t2a", C.HIFIGAN_16K, 624, manifest, mel=spec)
    hifigan_case("hifigan_ns512", C.HIFIGAN_NS_512, 64, manifest, B=2)
    hifigan_case("hifigan_ns128", C.HIFIGAN_NS_128, 96, manifest, B=2)
    bigvgan_case("bigvgan_16k", C.BIGVGAN_16K, 48, manifest)
    hifigan_case("hifigan_rb2", C.HIFIGAN_RB2, 72, manifest, B=2, seed=12)
    bigvgan_case("bigvgan_rb2", C.BIGVGAN_RB2, 40, manifest, seed=13)
    hifigan_nsf_case("hifigan_nsf_24k", C.HIFIGAN_NSF_24K, 40, manifest)
    diffsinger_case("diffsinger_ds1000", C.DIFFSINGER_DS1000, 48, manifest)
    config2_case("t2a_config2_s100")
    bigvgan_case("bigvgan_16k_t624", C.BIGVGAN_16K, 624, {})
    config3_case("hifigan_ns512_cfg3_row63", C.HIFIGAN_NS_512, 63)
    config3_case("hifigan_ns128_cfg3_row0", C.HIFIGAN_NS_128, 0)
    try:          # records that are not state_dict layouts ("_third_party_pins") survive a full regeneration
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest.update({k: v for k, v in json.load(f).items() if k.startswith("_")})
    except (OSError, ValueError):
        pass
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("torch", torch.__version__)


def main_diffsinger_only():
    """`python tests/golden/make_golden.py diffsinger`: add the DiffSinger case without regenerating the others."""
    torch.set_num_threads(8)
    _install_shims()
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    diffsinger_case("diffsinger_ds1000", C.DIFFSINGER_DS1000, 48, manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("torch", torch.__version__)


def main_nsf_only():
    """`python tests/golden/make_golden.py nsf`: add the NSF case without regenerating the other fixtures."""
    torch.set_num_threads(8)
    _install_shims()
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    hifigan_nsf_case("hifigan_nsf_24k", C.HIFIGAN_NSF_24K, 40, manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("torch", torch.__version__)


def main_resblock2_only():
    """`python tests/golden/make_golden.py resblock2`: the generators' other residual block (`resblock: "2"`) through the
    reference's own Generator / HifiGanGenerator (ResBlock2) and BigVGAN (AMPBlock2, plain `snake`)."""
    torch.set_num_threads(8)
    _install_shims()
    with open(os.path.join(HERE, "manifest.json")) as f:
        manifest = json.load(f)
    hifigan_case("hifigan_rb2", C.HIFIGAN_RB2, 72, manifest, B=2, seed=12)
    bigvgan_case("bigvgan_rb2", C.BIGVGAN_RB2, 40, manifest, seed=13)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print("torch", torch.__version__)


def main_ddim_full_only():
    """`python tests/golden/make_golden.py ddimfull`: the mask / eta / intermediates case of the sampler only."""
    _install_shims()
    u = unet_case("unet_t2a", C.UNET_T2A, 10, 78, 77, {}, save=False)
    ddim_full_signature_case("ddim_t2a_mask_eta_s6", u)


def main_ddim_hooks_only():
    """`python tests/golden/make_golden.py ddimhooks`: the score-corrector / callback case of the sampler only."""
    torch.set_num_threads(8)
    _install_shims()
    u = unet_case("unet_t2a", C.UNET_T2A, 10, 78, 77, {}, save=False)
    ddim_host_hooks_case("ddim_t2a_host_hooks_s4", u)


def main_ddim_variants_only():
    """`python tests/golden/make_golden.py ddimvar`: add the I2A / inpaint sampler cases only."""
    torch.set_num_threads(8)
    _install_shims()
    scratch = {}
    u_i2a = unet_case("unet_i2a", C.UNET_I2A, 10, 78, 1, scratch, seed=4, save=False)
    u_inp = unet_case("unet_inpaint", C.UNET_INPAINT, 10, 106, 0, scratch, n=1, seed=5, save=False)
    ddim_variant_case("ddim_i2a_s4", u_i2a, C.LDM_I2A, 4, 3.0, ctx_len=1)
    ddim_variant_case("ddim_inpaint_s4", u_inp, C.LDM_INPAINT, 4, 1.0)
    print("torch", torch.__version__)


if __name__ == "__main__":
    {"nsf": main_nsf_only, "ddimvar": main_ddim_variants_only, "ddimfull": main_ddim_full_only, "diffsinger": main_diffsinger_only,
     "config2": main_config2_only, "config3": main_config3_only, "encoders": main_encoders_only, "cliptext": main_clip_text_only, "clapaudio": main_clap_audio_only, "clapscore": main_clap_score_only, "mixed": main_mixed_only,
     "resblock2": main_resblock2_only, "ddimhooks": main_ddim_hooks_only}.get(" ".join(sys.argv[1:]), main)()
