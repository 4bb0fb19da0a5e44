"""Handle objects over the C ABI: context, UNet, VAE, vocoder, and the single-operator entry points.

torch is used for device buffers and the stream only; every computation below happens inside
libaudiogpt_mi355x.so.  All tensors at this boundary are fp32, contiguous, in the reference's layouts.
"""
import ctypes as C
import threading
import weakref

import numpy as np
import torch

from . import _lib as L


def default_precision():
    """Precision mode of the drop-in tool / sampler / vocoder classes when none is given: bf16x3 (meets the fp32
    parity gates), overridable with AUDIOGPT_AMD_PRECISION=f32|bf16x3|bf16."""
    import os
    p = os.environ.get("AUDIOGPT_AMD_PRECISION", "bf16x3")
    if p not in Context.PRECISIONS:
        raise ValueError("AUDIOGPT_AMD_PRECISION must be one of %s" % sorted(Context.PRECISIONS))
    return p


_contexts = weakref.WeakSet()


def reload_tuning():
    """Make every live context parse the MAA_* tuning environment again (it is read when a context is created): tests and
    A/B scripts call this after changing os.environ."""
    for c in list(_contexts):
        c.reload_tuning()


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Context:
    """One (device, stream) execution context.  Calls are serialised with a lock, like the reference's
    non re-entrant sampler (ddim.py:27-56)."""

    PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2}

    def __init__(self, device="cuda:0", stream=None, precision="f32"):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.MaaError("no MI355X visible: the HIP backend has no CPU fallback")
        self.device = torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        torch.cuda.set_device(idx)
        # stream=None: the library creates its own (blocking) HIP stream, which orders against PyTorch's
        # default stream and can be captured into a hipGraph; pass a torch.cuda.Stream to share one.
        self._stream = stream
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else None
        h = C.c_void_p()
        L.check(self.lib.maa_ctx_create(idx, sp, C.byref(h)))
        self.h = h
        self.lock = threading.RLock()
        self.precision = "f32"
        self.set_precision(precision)
        _contexts.add(self)

    def reload_tuning(self):
        if getattr(self, "h", None):
            with self.lock:
                L.check(self.lib.maa_ctx_reload_tuning(self.h))

    def set_precision(self, precision):
        """Arithmetic of the contractions for models created from now on (and for the op_* calls)."""
        L.check(self.lib.maa_ctx_set_precision(self.h, self.PRECISIONS[precision]))
        self.precision = precision

    def set_cfg_split(self, mode):
        """Classifier-free guidance inside `ddim_sample`: True -> the two halves of the UNet batch as two lanes (two branches of
        the captured step graph), False -> one stream, None -> the default policy: two lanes unless the context has been told that three or more are in
        flight on the device (`set_concurrency`; one batch owning the GPU gains 4 %, with three in flight the extra lanes lose up to 24 %).  The results are the same bit for bit."""
        L.check(self.lib.maa_ctx_set_cfg_split(self.h, -1 if mode is None else int(bool(mode))))

    def set_concurrency(self, n):
        """The serving arrangement as a hint: how many contexts' launches the caller keeps in flight on this device (None: not told,
        treated as 1).  >= 3: one stream per guided DDIM step and tiles by least total workgroup time; 1 or 2: this
        context (nearly) owns the GPU (two CFG lanes, tiles by least launch time).  Bit-identical either way (include/maa.h)."""
        L.check(self.lib.maa_ctx_set_concurrency(self.h, -1 if n is None else int(n)))

    def synchronize(self):
        L.check(self.lib.maa_ctx_synchronize(self.h))

    def workspace_bytes(self):
        n = C.c_size_t()
        L.check(self.lib.maa_ctx_workspace_bytes(self.h, C.byref(n)))
        return n.value

    def prof_begin(self, detail=False):
        """Start per-kernel hipEvent timing of every launch on this context (eager launches only)."""
        L.check(self.lib.maa_prof_begin(self.h, int(detail)))

    def prof_end(self):
        """Stop timing; returns {kernel: dict(launches, ms, flops, bytes)}."""
        rows = (L.maa_prof_row * 512)()
        n = C.c_int()
        L.check(self.lib.maa_prof_end(self.h, rows, 512, C.byref(n)))
        return {rows[i].name.decode(): dict(launches=int(rows[i].launches), ms=rows[i].ms, flops=rows[i].flops,
                                            bytes=rows[i].bytes) for i in range(n.value)}

    def calib(self):
        """Box calibration (csrc/calib.hip): what a fixed MFMA loop, a fixed device copy and fixed re-reads out of L2 / out of
        the Infinity Cache reach on this box right now."""
        out = {}
        for kind, key in ((0, "mfma_bf16_tflops"), (1, "copy_gbs"), (2, "l2_read_gbs"), (3, "infinity_cache_read_gbs")):
            v = C.c_double()
            L.check(self.lib.maa_calib(self.h, kind, C.byref(v)))
            out[key] = v.value
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.maa_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- single operators (parity tests / kernel profiling) -------------------------------------
    def op_linear(self, a, w, b=None, geglu=False):
        a = _f32(a, self.device)
        M, K = a.shape
        N = w.shape[0]
        wt, wp = L.host_f32(w)
        bt, bp = L.host_f32(b) if b is not None else (None, None)
        y = torch.empty(M, N // 2 if geglu else N, device=self.device)
        L.check(self.lib.maa_op_linear(self.h, L.dptr(a), M, K, wp, bp, N, int(geglu), L.dptr(y)))
        return y

    def op_conv(self, x, w, b=None, stride=1, pad=0, dil=1, up=False, leaky=0.0, out_hw=None):
        """x [B,Cin,H,W]; w [Cout,Cin,KH,KW]; returns [B,Cout,Ho,Wo]."""
        x = _f32(x, self.device)
        B, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        if out_hw is None:
            He, We = (2 * H, 2 * W) if up else (H, W)
            ph = pad if KH > 1 else 0
            Ho = (He + 2 * ph - dil * (KH - 1) - 1) // stride + 1
            Wo = (We + 2 * pad - dil * (KW - 1) - 1) // stride + 1
        else:
            Ho, Wo = out_hw
        wt, wp = L.host_f32(w)
        bt, bp = L.host_f32(b) if b is not None else (None, None)
        y = torch.empty(B, Cout, Ho, Wo, device=self.device)
        L.check(self.lib.maa_op_conv(self.h, L.dptr(x), B, Cin, H, W, wp, bp, Cout, KH, KW, stride, pad, dil,
                                     int(up), float(leaky), L.dptr(y), Ho, Wo))
        return y

    def op_groupnorm(self, x, gamma, beta, eps, silu=False):
        x = _f32(x, self.device)
        B, Cc = x.shape[:2]
        HW = int(np.prod(x.shape[2:]))
        gt, gp = L.host_f32(gamma)
        bt, bp = L.host_f32(beta)
        y = torch.empty_like(x)
        L.check(self.lib.maa_op_groupnorm(self.h, L.dptr(x), B, Cc, HW, gp, bp, float(eps), int(silu), L.dptr(y)))
        return y

    def op_layernorm(self, x, gamma, beta, eps=1e-5):
        x = _f32(x, self.device)
        rows, Cc = x.reshape(-1, x.shape[-1]).shape
        gt, gp = L.host_f32(gamma)
        bt, bp = L.host_f32(beta)
        y = torch.empty_like(x)
        L.check(self.lib.maa_op_layernorm(self.h, L.dptr(x), rows, Cc, gp, bp, float(eps), L.dptr(y)))
        return y

    def op_attention(self, q, k, v, heads, alpha):
        q, k, v = _f32(q, self.device), _f32(k, self.device), _f32(v, self.device)
        B, Nq, Cc = q.shape
        Nk = k.shape[1]
        y = torch.empty_like(q)
        L.check(self.lib.maa_op_attention(self.h, L.dptr(q), L.dptr(k), L.dptr(v), B, heads, Cc // heads, Nq, Nk,
                                          float(alpha), L.dptr(y)))
        return y

    def op_conv_transpose1d(self, x, w, b, stride, leaky=0.0):
        x = _f32(x, self.device)
        B, Cin, Ln = x.shape
        _, Cout, k = w.shape
        wt, wp = L.host_f32(w)
        bt, bp = L.host_f32(b)
        y = torch.empty(B, Cout, Ln * stride, device=self.device)
        L.check(self.lib.maa_op_conv_transpose1d(self.h, L.dptr(x), B, Cin, Ln, wp, bp, Cout, k, stride, float(leaky),
                                                 L.dptr(y)))
        return y

    def op_bench_conv(self, B, H, W, Cin, Cout, taps=9, pre_split=True, iters=20):
        """Kernel-only time (ms per launch) of one conv in this context's precision mode, synthetic data."""
        ms = C.c_float()
        L.check(self.lib.maa_op_bench_conv(self.h, B, H, W, Cin, Cout, taps, int(pre_split), iters, C.byref(ms)))
        return ms.value

    def op_snake_aa(self, x, alpha, beta, logscale):
        x = _f32(x, self.device)
        B, Cc, Ln = x.shape
        at, ap = L.host_f32(alpha)
        bt, bp = L.host_f32(beta)
        y = torch.empty_like(x)
        L.check(self.lib.maa_op_snake_aa(self.h, L.dptr(x), B, Cc, Ln, ap, bp, int(logscale), L.dptr(y)))
        return y


def _fill(arr, values):
    for i, v in enumerate(values):
        arr[i] = int(v)
    return len(values)


class UNet:
    """maa_unet handle: replaces instantiate_from_config(unet_config) + load_state_dict."""

    def __init__(self, ctx, cfg, state_dict):
        self.ctx, self.cfg = ctx, cfg
        c = L.maa_unet_config()
        c.in_channels, c.out_channels, c.model_channels = cfg["in_channels"], cfg["out_channels"], cfg["model_channels"]
        c.num_res_blocks = cfg["num_res_blocks"]
        c.n_channel_mult = _fill(c.channel_mult, cfg["channel_mult"])
        c.n_attention_resolutions = _fill(c.attention_resolutions, cfg["attention_resolutions"])
        c.num_heads, c.num_head_channels = cfg["num_heads"], cfg["num_head_channels"]
        c.use_spatial_transformer = int(cfg["use_spatial_transformer"])
        c.transformer_depth = cfg.get("transformer_depth", 1)
        c.context_dim = cfg["context_dim"] or 0
        c.legacy, c.resblock_updown = int(cfg["legacy"]), int(cfg["resblock_updown"])
        c.add_context_to_emb = int(cfg.get("add_context_to_emb", False))
        arr, n, keep = L.tensor_list(state_dict)
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_unet_create(ctx.h, C.byref(c), arr, n, C.byref(h)))
        self.h = h
        self._context = None

    def set_context(self, context):
        """context [B, L, context_dim] on the device; K/V projections are cached for the next forwards."""
        context = _f32(context, self.ctx.device)
        self._context = context                 # keep alive: the I2A variant re-reads it every forward
        B, Ln, _ = context.shape
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_unet_set_context(self.ctx.h, self.h, L.dptr(context), B, Ln))

    def forward(self, x, t, context=None):
        """UNetModel.forward(x, timesteps, context) (openaimodel.py:711-744)."""
        if context is not None:
            self.set_context(context)
        x = _f32(x, self.ctx.device)
        tf = t.to(device=self.ctx.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        out = torch.empty(B, self.cfg["out_channels"], H, W, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_unet_forward(self.ctx.h, self.h, L.dptr(x), L.dptr(tf), B, H, W, L.dptr(out)))
        return out

    __call__ = forward

    def ddim_sample(self, x_T, timesteps, alphas, alphas_prev, cond=None, uncond=None, scale=1.0, concat=None,
                    use_graph=True, mask=None, x0=None, noise_q=None, sqrt_ac=None, sqrt_1mac=None, sigmas=None,
                    noise_p=None, temperature=1.0, log_every_t=None):
        """Whole DDIM trajectory on the device (ddim.py:118-225).  Returns x_0, or (x_0, x_inter, pred_x0) when
        log_every_t is given (the two logs as [n_log, B, C, H, W] tensors, ddim.py:158-163).
        mask / x0 / noise_q [S, B, C, H, W] / sqrt_ac, sqrt_1mac [S]: the mask blend of ddim.py:147-150;
        sigmas [S] / noise_p [S, B, C, H, W] / temperature: the eta > 0 noise term of ddim.py:210-225.  Noise tensors
        are in loop order (first step first)."""
        dev = self.ctx.device
        x = _f32(x_T, dev).clone()
        B, Cc, H, W = x.shape
        a = L.maa_ddim_args()
        ts = np.ascontiguousarray(np.asarray(timesteps), dtype=np.int32)
        al = np.ascontiguousarray(np.asarray(alphas), dtype=np.float32)
        ap = np.ascontiguousarray(np.asarray(alphas_prev), dtype=np.float32)
        a.S, a.B, a.C, a.H, a.W = len(ts), B, Cc, H, W
        a.scale = float(scale)
        keep = []
        # shapes are checked here (the C ABI sees bare pointers): the reference raises from torch.cat / the attention
        # einsum on any of these mismatches (ddim.py:177-199, ddpm.py:1404-1406)
        if cond is not None:
            cond = _f32(cond, dev)
            cdim = self.cfg["context_dim"] or 0
            if cond.dim() != 3 or cond.shape[0] != B or cond.shape[2] != cdim:
                raise L.MaaError("ddim_sample: conditioning must be [B=%d, L, %d], got %s" % (B, cdim, tuple(cond.shape)))
            keep.append(cond)
            a.d_cond = cond.data_ptr()
            a.L = cond.shape[1]
        if uncond is not None:
            if cond is None:
                raise L.MaaError("ddim_sample: unconditional_conditioning without conditioning")
            uncond = _f32(uncond, dev)
            if tuple(uncond.shape) != tuple(cond.shape):
                raise L.MaaError("ddim_sample: unconditional_conditioning %s must have the shape of conditioning %s"
                                 % (tuple(uncond.shape), tuple(cond.shape)))
            keep.append(uncond)
            a.d_uncond = uncond.data_ptr()
        if concat is not None:
            concat = _f32(concat, dev)
            if concat.dim() != 4 or concat.shape[0] != B or tuple(concat.shape[2:]) != (H, W) \
                    or Cc + concat.shape[1] != self.cfg["in_channels"]:
                raise L.MaaError("ddim_sample: concat conditioning must be [B=%d, %d, %d, %d], got %s"
                                 % (B, self.cfg["in_channels"] - Cc, H, W, tuple(concat.shape)))
            keep.append(concat)
            a.d_concat = concat.data_ptr()
            a.Cc = concat.shape[1]
        a.h_timesteps = ts.ctypes.data_as(C.POINTER(C.c_int32))
        a.h_alphas = al.ctypes.data_as(C.POINTER(C.c_float))
        a.h_alphas_prev = ap.ctypes.data_as(C.POINTER(C.c_float))
        a.use_graph = int(use_graph)
        S = len(ts)

        def host_table(v, what):
            t = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            if t.shape != (S,):
                raise L.MaaError("ddim_sample: %s must hold one value per DDIM step (%d), got %s" % (what, S, t.shape))
            keep.append(t)
            return t.ctypes.data_as(C.POINTER(C.c_float))

        def step_noise(v, what):
            t = _f32(v, dev)
            if tuple(t.shape) != (S, B, Cc, H, W):
                raise L.MaaError("ddim_sample: %s must be [S=%d, %d, %d, %d, %d], got %s" % (what, S, B, Cc, H, W, tuple(t.shape)))
            keep.append(t)
            return t.data_ptr()

        if mask is not None:
            if x0 is None or noise_q is None or sqrt_ac is None or sqrt_1mac is None:
                raise L.MaaError("ddim_sample: mask needs x0, noise_q and the q_sample tables")      # ddim.py:148 asserts x0
            m = _f32(mask, dev).expand(B, Cc, H, W).contiguous()
            z0 = _f32(x0, dev).expand(B, Cc, H, W).contiguous()
            keep += [m, z0]
            a.d_mask, a.d_x0 = m.data_ptr(), z0.data_ptr()
            a.d_noise_q = step_noise(noise_q, "noise_q")
            a.h_sqrt_ac, a.h_sqrt_1mac = host_table(sqrt_ac, "sqrt_ac"), host_table(sqrt_1mac, "sqrt_1mac")
        if sigmas is not None:
            if noise_p is None:
                raise L.MaaError("ddim_sample: sigmas (eta > 0) need noise_p")
            a.h_sigmas = host_table(sigmas, "sigmas")
            a.d_noise_p = step_noise(noise_p, "noise_p")
        a.temperature = float(temperature)
        logs = None
        if log_every_t is not None:
            n_log = sum(1 for i in range(S) if i % int(log_every_t) == 0 or i == S - 1)
            # the log slabs are part of the captured step (their addresses are in the step graph's key, csrc/ddim.cpp): one pair
            # per shape is kept by this model and the caller gets copies, so that a second sample() call replays the kept graph
            # whatever the caching allocator would have handed out for fresh tensors
            cache = self.__dict__.setdefault("_log_slabs", {})
            key = (n_log, B, Cc, H, W)
            if key not in cache:
                if len(cache) >= 4:
                    cache.clear()
                cache[key] = (torch.empty(n_log, B, Cc, H, W, device=dev), torch.empty(n_log, B, Cc, H, W, device=dev))
            logs = cache[key]
            a.log_every_t, a.n_log = int(log_every_t), n_log
            a.d_log_x, a.d_log_x0 = logs[0].data_ptr(), logs[1].data_ptr()
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_ddim_sample(self.ctx.h, self.h, C.byref(a), L.dptr(x)))
            if logs is not None:
                # the slabs are shared by every call on this model: copy them out before the lock is released, ordered behind the
                # context's own stream (a private torch stream: copy on it; a library-created blocking stream orders against
                # torch's default stream by itself, but a caller on another stream must wait, so drain it)
                if self.ctx._stream is not None:
                    with torch.cuda.stream(self.ctx._stream):
                        out = (logs[0].clone(), logs[1].clone())
                    for t in out:
                        t.record_stream(torch.cuda.current_stream(dev))
                    torch.cuda.current_stream(dev).wait_stream(self.ctx._stream)
                else:
                    self.ctx.synchronize()
                    out = (logs[0].clone(), logs[1].clone())
                return x, out[0], out[1]
        return x

    def ddim_update(self, x, eps_uncond, eps_cond, scale, a_t, a_prev, sigma_t, sqrt_one_minus_at):
        """p_sample_ddim's elementwise tail for ONE step (ddim.py:199, 210-225 without the noise term) through
        `maa_ddim_update`: e = eu + scale (ec - eu) (eps_cond None: e = eps_uncond), returns (x_prev, pred_x0).  Used by the
        sampler's host-side loop (score correctors / callbacks); the device loop has its own fused form."""
        dev = self.ctx.device
        x, eu = _f32(x, dev), _f32(eps_uncond, dev)
        ec = None if eps_cond is None else _f32(eps_cond, dev)
        coef = torch.tensor([a_t, a_prev, sigma_t, sqrt_one_minus_at], dtype=torch.float32).to(dev)
        x_prev, pred = torch.empty_like(x), torch.empty_like(x)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_ddim_update(self.ctx.h, L.dptr(x), L.dptr(eu), L.dptr(ec) if ec is not None else None,
                                                 float(scale), L.dptr(coef), x.numel(), L.dptr(x_prev), L.dptr(pred)))
        return x_prev, pred

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_unet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VAE:
    def __init__(self, ctx, dd, state_dict):
        self.ctx, self.dd = ctx, dd
        c = L.maa_vae_config()
        c.ch, c.out_ch, c.in_channels, c.z_channels = dd["ch"], dd["out_ch"], dd["in_channels"], dd["z_channels"]
        c.embed_dim, c.resolution, c.num_res_blocks = dd["embed_dim"], dd["resolution"], dd["num_res_blocks"]
        c.double_z = int(dd["double_z"])
        c.n_ch_mult = _fill(c.ch_mult, dd["ch_mult"])
        c.n_attn_resolutions = _fill(c.attn_resolutions, dd["attn_resolutions"])
        arr, n, keep = L.tensor_list(state_dict)
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_vae_create(ctx.h, C.byref(c), arr, n, C.byref(h)))
        self.h = h

    def decode(self, z, scale_factor=1.0):
        """decode_first_stage (ddpm_audio.py:352-359): z [B,4,h,w] -> mel [B,1,8h,8w]."""
        z = _f32(z, self.ctx.device)
        B, _, h, w = z.shape
        f = 2 ** (len(self.dd["ch_mult"]) - 1)
        mel = torch.empty(B, self.dd["out_ch"], h * f, w * f, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_vae_decode(self.ctx.h, self.h, L.dptr(z), B, h, w, 1.0 / float(scale_factor),
                                                L.dptr(mel)))
        return mel

    def decode_spec(self, z, scale_factor=1.0):
        """decode_first_stage + the tools' clamp((x + 1) / 2, 0, 1) in the decoder's last pass: z [B,4,h,w] -> spec [B,8h,8w]."""
        z = _f32(z, self.ctx.device)
        B, _, h, w = z.shape
        f = 2 ** (len(self.dd["ch_mult"]) - 1)
        spec = torch.empty(B, h * f, w * f, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_vae_decode_spec(self.ctx.h, self.h, L.dptr(z), B, h, w, 1.0 / float(scale_factor),
                                                     L.dptr(spec)))
        return spec

    def encode_moments(self, mel):
        """AutoencoderKL.encode moments (autoencoder.py:345-349): mel [B,1,H,W] -> [B, 2*embed, H/8, W/8]."""
        mel = _f32(mel, self.ctx.device)
        B, _, H, W = mel.shape
        f = 2 ** (len(self.dd["ch_mult"]) - 1)
        out = torch.empty(B, 2 * self.dd["embed_dim"], H // f, W // f, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_vae_encode_moments(self.ctx.h, self.h, L.dptr(mel), B, H, W, L.dptr(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_vae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Vocoder:
    def __init__(self, ctx, cfg, state_dict):
        self.ctx, self.cfg = ctx, cfg
        c = L.maa_vocoder_config()
        c.kind = 1 if cfg["kind"] == "bigvgan" else 0
        c.num_mels, c.upsample_initial_channel = cfg["num_mels"], cfg["upsample_initial_channel"]
        c.n_upsamples = _fill(c.upsample_rates, cfg["upsample_rates"])
        _fill(c.upsample_kernel_sizes, cfg["upsample_kernel_sizes"])
        c.n_kernels = _fill(c.resblock_kernel_sizes, cfg["resblock_kernel_sizes"])
        c.n_dilations = len(cfg["resblock_dilation_sizes"][0])
        for j, ds in enumerate(cfg["resblock_dilation_sizes"]):
            for m, d in enumerate(ds):
                c.resblock_dilation_sizes[j][m] = int(d)
        c.snake_beta = int(cfg.get("activation", "") == "snakebeta")
        c.snake_logscale = int(cfg.get("snake_logscale", False))
        self.nsf = bool(cfg.get("use_pitch_embed", False))
        c.use_pitch_embed = int(self.nsf)
        c.sampling_rate = int(cfg.get("sampling_rate", 0))
        c.harmonic_num = self.harmonics = 8 if self.nsf else 0            # hifigan.py:112
        c.resblock = int(cfg.get("resblock", "1"))                        # hifigan.py:119: '1' -> ResBlock1, else ResBlock2
        if c.resblock != 1:
            c.resblock = 2
        arr, n, keep = L.tensor_list(state_dict)
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_vocoder_create(ctx.h, C.byref(c), arr, n, C.byref(h)))
        self.h = h
        self.hop = int(np.prod(cfg["upsample_rates"]))

    def forward(self, mel):
        """mel [B, num_mels, T] -> wav [B, 1, T*hop] (HifiGanGenerator.forward, hifigan.py:144-169)."""
        mel = _f32(mel, self.ctx.device)
        B, _, T = mel.shape
        wav = torch.empty(B, 1, T * self.hop, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_vocoder_forward(self.ctx.h, self.h, L.dptr(mel), B, T, L.dptr(wav)))
        return wav

    def forward_f0(self, mel, f0, rand_ini=None, noise=None):
        """NSF branch (HifiGanGenerator.forward(x, f0), hifigan.py:144-157): mel [B, num_mels, T], f0 [B, T] in Hz.
        rand_ini [B, 9] / noise [B, T*hop, 9] are the two tensors SineGen.forward draws (source.py:355-358, 425); when
        omitted they are drawn here with torch's global generator in the reference's order and on the mel's device."""
        if not self.nsf:
            raise L.MaaError("this generator has no NSF branch (use_pitch_embed is off)")
        dev = self.ctx.device
        mel, f0 = _f32(mel, dev), _f32(f0, dev)
        B, _, T = mel.shape
        if tuple(f0.shape) != (B, T):
            raise L.MaaError("forward_f0: f0 must be [B=%d, T=%d], got %s" % (B, T, tuple(f0.shape)))
        H1, Ln = self.harmonics + 1, T * self.hop
        if rand_ini is None:
            rand_ini = torch.rand(B, H1, device=dev)
        if noise is None:
            noise = torch.randn(B, Ln, H1, device=dev)
        rand_ini, noise = _f32(rand_ini, dev), _f32(noise, dev)
        if tuple(rand_ini.shape) != (B, H1) or tuple(noise.shape) != (B, Ln, H1):
            raise L.MaaError("forward_f0: rand_ini must be [%d, %d] and noise [%d, %d, %d]" % (B, H1, B, Ln, H1))
        wav = torch.empty(B, 1, Ln, device=dev)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_vocoder_forward_f0(self.ctx.h, self.h, L.dptr(mel), L.dptr(f0), L.dptr(rand_ini),
                                                        L.dptr(noise), B, T, L.dptr(wav)))
        return wav

    def __call__(self, mel, f0=None, **kw):
        return self.forward(mel) if f0 is None else self.forward_f0(mel, f0, **kw)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_vocoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DiffNet:
    """maa_diffnet handle: DiffSinger's denoiser (NeuralSeq/modules/diff/net.py:84-130) and the PLMS loop over it
    (NeuralSeq/modules/diff/shallow_diffusion_tts.py:166-201, 262-269)."""

    def __init__(self, ctx, cfg, state_dict):
        self.ctx, self.cfg = ctx, cfg
        c = L.maa_diffnet_config()
        c.in_dims, c.hidden_size = cfg["in_dims"], cfg["hidden_size"]
        c.residual_layers, c.residual_channels = cfg["residual_layers"], cfg["residual_channels"]
        c.dilation_cycle_length = cfg["dilation_cycle_length"]
        arr, n, keep = L.tensor_list(state_dict)
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_diffnet_create(ctx.h, C.byref(c), arr, n, C.byref(h)))
        self.h = h

    def forward(self, spec, diffusion_step, cond):
        """spec [B, 1, M, T], diffusion_step [B] or [B, 1] (integer steps), cond [B, H, T] -> eps [B, 1, M, T]."""
        dev = self.ctx.device
        spec, cond = _f32(spec, dev), _f32(cond, dev)
        t = _f32(torch.as_tensor(diffusion_step).reshape(-1), dev)
        B, _, M, T = spec.shape
        if M != self.cfg["in_dims"] or tuple(cond.shape) != (B, self.cfg["hidden_size"], T) or t.shape[0] != B:
            raise L.MaaError("DiffNet.forward: spec %s / step %s / cond %s do not fit in_dims %d, hidden_size %d"
                             % (tuple(spec.shape), tuple(t.shape), tuple(cond.shape), self.cfg["in_dims"], self.cfg["hidden_size"]))
        out = torch.empty_like(spec)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_diffnet_forward(self.ctx.h, self.h, L.dptr(spec), L.dptr(t), L.dptr(cond), B, T, L.dptr(out)))
        return out

    __call__ = forward

    def plms_sample(self, x, cond, alphas_cumprod, K_step, interval, use_graph=True):
        """x [B, 1, M, T] = x_K -> x_0 after the pndm_speedup loop: t = K_step - interval, ..., 0."""
        dev = self.ctx.device
        x = _f32(x, dev).clone()
        cond = _f32(cond, dev)
        B, _, M, T = x.shape
        if M != self.cfg["in_dims"] or tuple(cond.shape) != (B, self.cfg["hidden_size"], T):
            raise L.MaaError("plms_sample: x %s / cond %s do not fit the denoiser" % (tuple(x.shape), tuple(cond.shape)))
        if B > 256:
            raise L.MaaError("plms_sample: at most 256 samples per call (got %d)" % B)
        ac = np.ascontiguousarray(np.asarray(alphas_cumprod), dtype=np.float32)
        a = L.maa_plms_args()
        a.B, a.T, a.K_step, a.interval, a.timesteps = B, T, int(K_step), int(interval), int(ac.shape[0])
        a.d_cond = cond.data_ptr()
        a.h_alphas_cumprod = ac.ctypes.data_as(C.POINTER(C.c_float))
        a.use_graph = int(use_graph)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_plms_sample(self.ctx.h, self.h, C.byref(a), L.dptr(x)))
        return x

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_diffnet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Encoder:
    """maa_encoder handle: a conditioning tower on the device.

    kind "text":  BertModel(input_ids) + CLAP Projection per token, FrozenCLAPEmbedder.encode
                  (ldm/modules/encoders/modules.py:204-211, CLAP/clap.py:8-20)
    kind "image": open_clip VisionTransformer + L2 normalisation, FrozenGlobalNormOpenCLIPEmbedder.forward_img
                  (ldm/modules/encoders/modules.py:340-343)"""

    def __init__(self, ctx, cfg, state_dict):
        self.ctx, self.cfg = ctx, cfg
        c = L.maa_encoder_config()
        c.kind = {"text": 0, "image": 1, "clip_text": 2}[cfg["kind"]]
        c.layers, c.width, c.heads, c.mlp_dim, c.d_proj = cfg["layers"], cfg["width"], cfg["heads"], cfg["mlp_dim"], cfg["d_proj"]
        c.vocab, c.max_positions = cfg.get("vocab", 0), cfg.get("max_positions", 0)
        c.patch, c.image = cfg.get("patch", 0), cfg.get("image", 0)
        c.ln_eps = cfg["ln_eps"]
        arr, n, keep = L.tensor_list(state_dict)
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_encoder_create(ctx.h, C.byref(c), arr, n, C.byref(h)))
        self.h = h

    def encode_tokens(self, input_ids):
        """input_ids [B, L] (any integer dtype) -> [B, L, d_proj] (kind "text") / unit-length [B, d_proj] ("clip_text")."""
        if self.cfg["kind"] == "image":
            raise L.MaaError("encode_tokens on an image tower")
        ids = torch.as_tensor(input_ids)
        if ids.dim() != 2 or ids.shape[1] > self.cfg["max_positions"]:
            raise L.MaaError("encode_tokens: input_ids %s must be [B, L <= %d]" % (tuple(ids.shape), self.cfg["max_positions"]))
        ids = ids.to(device=self.ctx.device, dtype=torch.int32).contiguous()
        B, Ln = ids.shape
        shape = (B, Ln, self.cfg["d_proj"]) if self.cfg["kind"] == "text" else (B, self.cfg["d_proj"])
        out = torch.empty(*shape, dtype=torch.float32, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_encoder_text(self.ctx.h, self.h, C.c_void_p(ids.data_ptr()), B, Ln, L.dptr(out)))
        return out

    def encode_cls(self, input_ids):
        """The scorer's text side (wav_evaluation/models/clap.py:49-53 + CLAPWrapper.py:177-182): unpadded input_ids
        [B, L] -> Projection of the [CLS] row, unit length, [B, d_proj]."""
        if self.cfg["kind"] != "text":
            raise L.MaaError("encode_cls is the CLAP (BERT) tower's")
        ids = torch.as_tensor(input_ids)
        if ids.dim() != 2 or ids.shape[1] > self.cfg["max_positions"]:
            raise L.MaaError("encode_cls: input_ids %s must be [B, L <= %d]" % (tuple(ids.shape), self.cfg["max_positions"]))
        ids = ids.to(device=self.ctx.device, dtype=torch.int32).contiguous()
        out = torch.empty(ids.shape[0], self.cfg["d_proj"], dtype=torch.float32, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_encoder_text_cls(self.ctx.h, self.h, C.c_void_p(ids.data_ptr()), ids.shape[0],
                                                      ids.shape[1], L.dptr(out)))
        return out

    def encode_image(self, image):
        """image [B, 3, S, S] (preprocessed) -> [B, d_proj], rows of unit length."""
        if self.cfg["kind"] != "image":
            raise L.MaaError("encode_image on a text tower")
        x = _f32(image, self.ctx.device)
        S = self.cfg["image"]
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, S, S):
            raise L.MaaError("encode_image: image %s must be [B, 3, %d, %d]" % (tuple(x.shape), S, S))
        out = torch.empty(x.shape[0], self.cfg["d_proj"], dtype=torch.float32, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_encoder_image(self.ctx.h, self.h, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_encoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ClapAudio:
    """maa_clap_audio handle: Cnn14 from the log-mel on + Projection, unit-length rows (the audio side of the CLAP
    best-of-n scorer: wav_evaluation/models/audio.py:150-176, clap.py:8-39, CLAPWrapper.py:184-189)."""

    def __init__(self, ctx, cfg, state_dict):
        self.ctx, self.cfg = ctx, cfg
        c = L.maa_clap_audio_config()
        c.mel_bins, c.out_emb, c.d_proj, c.bn_eps = cfg["mel_bins"], cfg["out_emb"], cfg["d_proj"], cfg.get("bn_eps", 1e-5)
        c.n_blocks = _fill(c.channels, cfg["channels"])
        arr, n, keep = L.tensor_list({k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")})
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_clap_audio_create(ctx.h, C.byref(c), arr, n, C.byref(h)))
        self.h = h

    def embed(self, logmel, return_embedding=False):
        """logmel [B, 1, T, mel_bins] -> z [B, d_proj] (unit length) [, relu(fc1) embedding [B, out_emb]]."""
        x = _f32(logmel, self.ctx.device)
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[3] != self.cfg["mel_bins"]:
            raise L.MaaError("ClapAudio.embed: logmel %s must be [B, 1, T, %d]" % (tuple(x.shape), self.cfg["mel_bins"]))
        B, _, T, _ = x.shape
        z = torch.empty(B, self.cfg["d_proj"], dtype=torch.float32, device=self.ctx.device)
        emb = torch.empty(B, self.cfg["out_emb"], dtype=torch.float32, device=self.ctx.device) if return_embedding else None
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_clap_audio_embed(self.ctx.h, self.h, L.dptr(x), B, T,
                                                      L.dptr(emb) if emb is not None else None, L.dptr(z)))
        return (z, emb) if return_embedding else z

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_clap_audio_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def clap_similarity(ctx, audio_embeddings, text_embeddings, scale=1.0):
    """CLAPWrapper.compute_similarity (CLAPWrapper.py:207-215): [Na, D], [Nt, D] -> [Na, Nt] = scale * audio @ text^T."""
    a, t = _f32(audio_embeddings, ctx.device), _f32(text_embeddings, ctx.device)
    if a.dim() != 2 or t.dim() != 2 or a.shape[1] != t.shape[1]:
        raise L.MaaError("clap_similarity: embeddings %s / %s must be [N, D] with one D" % (tuple(a.shape), tuple(t.shape)))
    out = torch.empty(a.shape[0], t.shape[0], dtype=torch.float32, device=ctx.device)
    with ctx.lock:
        L.check(ctx.lib.maa_clap_similarity(ctx.h, L.dptr(a), L.dptr(t), a.shape[0], t.shape[0], a.shape[1], float(scale),
                                            L.dptr(out)))
    return out


class Spectral:
    """maa_spectral handle: waveform [B, n] -> log-mel (framed DFT GEMM -> |.|^p -> mel GEMM -> log), exact fp32.
    cfg: n_fft, hop, n_mels, pad_mode ("constant" | "reflect"), power (1 | 2), log_kind ("db" | "transforms_16000"),
    amin, ref, out_layout ("btm" = [B, frames, n_mels] | "bmt" = [B, n_mels, frames]); basis [2 n_freq, n_fft], melw
    [n_mels, n_freq] are host matrices (audiogpt_amd/mel.py builds them)."""

    def __init__(self, ctx, cfg, basis, melw):
        self.ctx, self.cfg = ctx, dict(cfg)
        c = L.maa_spectral_config()
        c.n_fft, c.hop, c.n_freq, c.n_mels = cfg["n_fft"], cfg["hop"], cfg["n_fft"] // 2 + 1, cfg["n_mels"]
        c.pad_mode = {"constant": 0, "reflect": 1}[cfg["pad_mode"]]
        c.power = int(cfg["power"])
        c.log_kind = {"db": 0, "transforms_16000": 1}[cfg["log_kind"]]
        c.amin, c.ref = float(cfg["amin"]), float(cfg.get("ref", 1.0))
        c.out_layout = {"btm": 0, "bmt": 1}[cfg["out_layout"]]
        bt, bp = L.host_f32(torch.as_tensor(basis))
        mt, mp = L.host_f32(torch.as_tensor(melw))
        if tuple(bt.shape) != (2 * c.n_freq, c.n_fft) or tuple(mt.shape) != (c.n_mels, c.n_freq):
            raise L.MaaError("Spectral: basis %s / melw %s do not match the configuration" % (tuple(bt.shape), tuple(mt.shape)))
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_spectral_create(ctx.h, C.byref(c), bp, mp, C.byref(h)))
        self.h = h

    def frames(self, n):
        return 1 + n // self.cfg["hop"]

    def forward(self, wav):
        x = _f32(wav, self.ctx.device)
        if x.dim() == 1:
            x = x[None]
        B, n = x.shape
        T, M = self.frames(n), self.cfg["n_mels"]
        out = torch.empty((B, T, M) if self.cfg["out_layout"] == "btm" else (B, M, T), dtype=torch.float32,
                          device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_spectral_forward(self.ctx.h, self.h, L.dptr(x), B, n, L.dptr(out)))
        return out

    __call__ = forward

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_spectral_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Resampler:
    """maa_resampler handle: torchaudio.transforms.Resample(orig, new) with its default sinc / Hann kernel bank
    (kernels [new/g, 2 width + orig/g], built on the host by audiogpt_amd/clap.sinc_resample_kernel), exact fp32."""

    def __init__(self, ctx, orig, new, width, kernels):
        self.ctx = ctx
        kt, kp = L.host_f32(torch.as_tensor(kernels))
        self.orig, self.new, self.width = int(orig), int(new), int(width)
        if kt.dim() != 2 or kt.shape[0] != self.new or kt.shape[1] != 2 * width + self.orig:
            raise L.MaaError("Resampler: kernel bank %s must be [new, 2 width + orig]" % (tuple(kt.shape),))
        h = C.c_void_p()
        with ctx.lock:
            L.check(ctx.lib.maa_resampler_create(ctx.h, self.orig, self.new, int(width), kt.shape[1], kp, C.byref(h)))
        self.h = h

    def forward(self, wav):
        x = _f32(wav, self.ctx.device)
        if x.dim() == 1:
            x = x[None]
        B, n = x.shape
        out = torch.empty(B, -(-self.new * n // self.orig), dtype=torch.float32, device=self.ctx.device)
        with self.ctx.lock:
            L.check(self.ctx.lib.maa_resampler_forward(self.ctx.h, self.h, L.dptr(x), B, n, L.dptr(out)))
        return out

    __call__ = forward

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.maa_resampler_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
