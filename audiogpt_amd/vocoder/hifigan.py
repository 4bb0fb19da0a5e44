"""HiFi-GAN / BigVGAN vocoder objects with the reference's Python surfaces, backed by the HIP library.

  HifiGanGenerator(h)(x[B,80,T], f0=None) -> [B,1,T*hop], .remove_weight_norm(), .load_state_dict(sd)
        NeuralSeq/modules/hifigan/hifigan.py:104-178, including the NSF branch (use_pitch_embed: f0 -> harmonic source ->
        noise_convs), which runs on the device too (csrc/nsf.hip)
  VocoderHifigan(ckpt_dir, device).vocode(spec)      text_to_audio/Make_An_Audio/vocoder/hifigan/modules.py:296-317
  VocoderBigVGAN(ckpt_dir, device).vocode(spec)      text_to_audio/Make_An_Audio/vocoder/bigvgan/models.py:393-414
  HifiGAN().spec2wav(mel[T,80]) + register_vocoder   NeuralSeq/vocoders/hifigan.py:39-69, vocoders/base_vocoder.py:1-39
`vocode` keeps the reference contract: Tensor [1,80,T] or ndarray [80,T] in, float32 ndarray [T*hop] out.
"""
import os

import numpy as np
import torch
import yaml

from .. import config as C
from .. import weights as WT
from ..backend import Context, Vocoder, default_precision

VOCODERS = {}


def register_vocoder(cls):
    """NeuralSeq/vocoders/base_vocoder.py:4-8."""
    VOCODERS[cls.__name__.lower()] = cls
    VOCODERS[cls.__name__] = cls
    return cls


def get_vocoder_cls(hparams):
    """base_vocoder.py:11-19: dotted path or registered name in hparams['vocoder']."""
    name = hparams["vocoder"]
    if name in VOCODERS:
        return VOCODERS[name]
    import importlib
    pkg, cls_name = name.rsplit(".", 1)
    return getattr(importlib.import_module(pkg), cls_name)


def _cfg_from_h(h, kind="hifigan"):
    g = (lambda k, d=None: h[k] if k in h else d) if isinstance(h, dict) else (lambda k, d=None: getattr(h, k, d))
    cfg = dict(kind=kind, num_mels=g("num_mels", g("audio_num_mel_bins", 80)),
               upsample_initial_channel=g("upsample_initial_channel"),
               upsample_rates=tuple(g("upsample_rates")), upsample_kernel_sizes=tuple(g("upsample_kernel_sizes")),
               resblock=str(g("resblock", "1")), resblock_kernel_sizes=tuple(g("resblock_kernel_sizes")),
               resblock_dilation_sizes=tuple(tuple(d) for d in g("resblock_dilation_sizes")),
               sampling_rate=g("sampling_rate", g("audio_sample_rate", 22050)))
    if g("use_pitch_embed", False):
        cfg["use_pitch_embed"] = True
        cfg["sampling_rate"] = g("audio_sample_rate", cfg["sampling_rate"])
    if kind == "bigvgan":
        cfg["activation"] = g("activation", "snakebeta")
        cfg["snake_logscale"] = bool(g("snake_logscale", True))
    return cfg


class HifiGanGenerator(object):
    """Callable generator; weights arrive through load_state_dict (reference key layout, with or without
    weight-norm) or are seeded-random until then."""

    def __init__(self, h, c_out=1, device="cuda:0", ctx=None, seed=2, precision=None):
        self.h = h
        self.cfg = _cfg_from_h(h)
        self.ctx = ctx or Context(device, precision=precision or default_precision())
        self.device = self.ctx.device
        self._sd = WT.make_vocoder_state_dict(self.cfg, seed=seed)
        self._impl = None

    def _build(self):
        if self._impl is None:
            self._impl = Vocoder(self.ctx, self.cfg, self._sd)
        return self._impl

    def load_state_dict(self, state, strict=True):
        if strict:
            want = set(WT.fold_weight_norm(self._sd).keys())
            got = set(WT.fold_weight_norm(dict(state)).keys())
            if want != got:
                raise RuntimeError("state_dict mismatch: missing %s unexpected %s" % (sorted(want - got)[:5], sorted(got - want)[:5]))
        self._sd = dict(state)
        if self._impl is not None:
            self._impl.close()
            self._impl = None

    def remove_weight_norm(self):
        """Weight-norm is always folded at load (numerically identical); kept for call compatibility."""
        return self

    def eval(self):
        return self

    def to(self, device):
        return self

    def forward(self, x, f0=None, rand_ini=None, noise=None):
        """hifigan.py:144-169.  f0 [B, T] engages the NSF branch (the generator must have been built with
        use_pitch_embed); rand_ini / noise optionally pin SineGen's two random draws."""
        if f0 is None:
            return self._build()(x)
        return self._build().forward_f0(x, f0, rand_ini=rand_ini, noise=noise)

    __call__ = forward


class _VocodeMixin(object):
    def vocode(self, spec, global_step=None):
        with torch.no_grad():
            if isinstance(spec, np.ndarray):
                spec = torch.from_numpy(spec).unsqueeze(0)
            spec = spec.to(dtype=torch.float32)
            if spec.dim() == 2:
                spec = spec.unsqueeze(0)
            return self.generator(spec).squeeze().cpu().numpy()

    def __call__(self, wav):
        return self.vocode(wav)


class _Gen(object):
    def __init__(self, ctx, cfg, sd):
        self.impl = Vocoder(ctx, cfg, sd)

    def __call__(self, x):
        return self.impl(x)


class VocoderHifigan(_VocodeMixin):
    def __init__(self, ckpt_vocoder=None, device="cuda:0", ctx=None, args=None, seed=2, precision=None):
        if args is None and ckpt_vocoder is not None and os.path.exists(os.path.join(ckpt_vocoder, "args.yml")):
            with open(os.path.join(ckpt_vocoder, "args.yml")) as f:
                args = yaml.safe_load(f)
        self.cfg = _cfg_from_h(args, "hifigan") if args is not None else dict(C.HIFIGAN_16K)
        self.ctx = ctx or Context(device, precision=precision or default_precision())
        self.device = self.ctx.device
        sd = None
        path = os.path.join(ckpt_vocoder, "best_netG.pt") if ckpt_vocoder else None
        if path and os.path.exists(path):
            sd = torch.load(path, map_location="cpu")["generator"]
        if sd is None:          # VocoderHifigan_noload behaviour (modules.py:319-332): run without a checkpoint
            sd = WT.make_vocoder_state_dict(self.cfg, seed=seed)
        self.generator = _Gen(self.ctx, self.cfg, sd)


class VocoderBigVGAN(_VocodeMixin):
    def __init__(self, ckpt_vocoder=None, device="cuda:0", ctx=None, args=None, seed=3, precision=None):
        if args is None and ckpt_vocoder is not None and os.path.exists(os.path.join(ckpt_vocoder, "args.yml")):
            with open(os.path.join(ckpt_vocoder, "args.yml")) as f:
                args = yaml.safe_load(f)
        self.cfg = _cfg_from_h(args, "bigvgan") if args is not None else dict(C.BIGVGAN_16K)
        self.ctx = ctx or Context(device, precision=precision or default_precision())
        self.device = self.ctx.device
        sd = None
        path = os.path.join(ckpt_vocoder, "best_netG.pt") if ckpt_vocoder else None
        if path and os.path.exists(path):
            sd = {k: v for k, v in torch.load(path, map_location="cpu")["generator"].items() if not k.endswith("filter")}
        if sd is None:
            sd = WT.make_vocoder_state_dict(self.cfg, seed=seed)
        self.generator = _Gen(self.ctx, self.cfg, sd)


class BaseVocoder(object):
    """NeuralSeq/vocoders/base_vocoder.py:22-39."""

    def spec2wav(self, mel):
        raise NotImplementedError


@register_vocoder
class HifiGAN(BaseVocoder):
    """NeuralSeq/vocoders/hifigan.py:39-69: spec2wav(mel [T,80]) -> wav [T*hop] float32 ndarray."""

    def __init__(self, hparams=None, device="cuda:0", ctx=None, state_dict=None, precision=None):
        h = dict(hparams or C.HIFIGAN_NS_512)
        h.setdefault("use_pitch_embed", False)
        self.use_nsf = bool(h.get("use_nsf", False))      # hparams.get('use_nsf') (vocoders/hifigan.py:60): falsy when absent
        self.model = HifiGanGenerator(h, device=device, ctx=ctx, precision=precision)
        if state_dict is not None:
            self.model.load_state_dict(state_dict, strict=True)
        self.device = self.model.device

    def spec2wav(self, mel, **kwargs):
        """vocoders/hifigan.py:55-69: f0 given and hparams['use_nsf'] -> self.model(c, f0), else self.model(c)."""
        with torch.no_grad():
            c = torch.as_tensor(mel, dtype=torch.float32).unsqueeze(0).transpose(2, 1)
            f0 = kwargs.get("f0")
            if f0 is not None and self.use_nsf:
                f0 = torch.as_tensor(np.asarray(f0), dtype=torch.float32)[None, :]
                y = self.model(c, f0).view(-1)
            else:
                y = self.model(c).view(-1)
        return y.cpu().numpy()
