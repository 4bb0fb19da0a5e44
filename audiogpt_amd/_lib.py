"""ctypes binding of libaudiogpt_mi355x.so (include/maa.h).  This is the binding a maintainer of the
reference would add; PyTorch is only used for device memory and the current stream.

There is no CPU fallback: if the shared object is missing or fails to load, importing the backend raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# AUDIOGPT_AMD_LIB: another build of the same library (same-box A/B of two builds; never a different implementation)
LIB_PATH = os.environ.get("AUDIOGPT_AMD_LIB") or os.path.join(_HERE, "libaudiogpt_mi355x.so")


class MaaError(RuntimeError):
    pass


class maa_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int), ("shape", C.c_int64 * 6)]


class maa_prof_row(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class maa_unet_config(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("model_channels", C.c_int),
                ("num_res_blocks", C.c_int),
                ("n_channel_mult", C.c_int), ("channel_mult", C.c_int * 8),
                ("n_attention_resolutions", C.c_int), ("attention_resolutions", C.c_int * 8),
                ("num_heads", C.c_int), ("num_head_channels", C.c_int),
                ("use_spatial_transformer", C.c_int), ("transformer_depth", C.c_int), ("context_dim", C.c_int),
                ("legacy", C.c_int), ("resblock_updown", C.c_int), ("add_context_to_emb", C.c_int)]


class maa_ddim_args(C.Structure):
    _fields_ = [("S", C.c_int), ("B", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("scale", C.c_float),
                ("d_cond", C.c_void_p), ("d_uncond", C.c_void_p), ("L", C.c_int),
                ("d_concat", C.c_void_p), ("Cc", C.c_int),
                ("h_timesteps", C.POINTER(C.c_int32)), ("h_alphas", C.POINTER(C.c_float)),
                ("h_alphas_prev", C.POINTER(C.c_float)), ("use_graph", C.c_int),
                ("d_mask", C.c_void_p), ("d_x0", C.c_void_p), ("d_noise_q", C.c_void_p),
                ("h_sqrt_ac", C.POINTER(C.c_float)), ("h_sqrt_1mac", C.POINTER(C.c_float)),
                ("h_sigmas", C.POINTER(C.c_float)), ("d_noise_p", C.c_void_p), ("temperature", C.c_float),
                ("log_every_t", C.c_int), ("n_log", C.c_int), ("d_log_x", C.c_void_p), ("d_log_x0", C.c_void_p)]


class maa_vae_config(C.Structure):
    _fields_ = [("ch", C.c_int), ("out_ch", C.c_int), ("in_channels", C.c_int), ("z_channels", C.c_int),
                ("embed_dim", C.c_int), ("resolution", C.c_int), ("num_res_blocks", C.c_int), ("double_z", C.c_int),
                ("n_ch_mult", C.c_int), ("ch_mult", C.c_int * 8),
                ("n_attn_resolutions", C.c_int), ("attn_resolutions", C.c_int * 8)]


class maa_vocoder_config(C.Structure):
    _fields_ = [("kind", C.c_int), ("num_mels", C.c_int), ("upsample_initial_channel", C.c_int),
                ("n_upsamples", C.c_int), ("upsample_rates", C.c_int * 8), ("upsample_kernel_sizes", C.c_int * 8),
                ("n_kernels", C.c_int), ("resblock_kernel_sizes", C.c_int * 8),
                ("n_dilations", C.c_int), ("resblock_dilation_sizes", (C.c_int * 8) * 8),
                ("snake_beta", C.c_int), ("snake_logscale", C.c_int),
                ("use_pitch_embed", C.c_int), ("sampling_rate", C.c_int), ("harmonic_num", C.c_int),
                ("resblock", C.c_int)]


class maa_diffnet_config(C.Structure):
    _fields_ = [("in_dims", C.c_int), ("hidden_size", C.c_int), ("residual_layers", C.c_int),
                ("residual_channels", C.c_int), ("dilation_cycle_length", C.c_int)]


class maa_encoder_config(C.Structure):
    _fields_ = [("kind", C.c_int), ("layers", C.c_int), ("width", C.c_int), ("heads", C.c_int), ("mlp_dim", C.c_int),
                ("d_proj", C.c_int), ("vocab", C.c_int), ("max_positions", C.c_int), ("patch", C.c_int),
                ("image", C.c_int), ("ln_eps", C.c_float)]


class maa_clap_audio_config(C.Structure):
    _fields_ = [("mel_bins", C.c_int), ("n_blocks", C.c_int), ("channels", C.c_int * 8), ("out_emb", C.c_int),
                ("d_proj", C.c_int), ("bn_eps", C.c_float)]


class maa_spectral_config(C.Structure):
    _fields_ = [("n_fft", C.c_int), ("hop", C.c_int), ("n_freq", C.c_int), ("n_mels", C.c_int), ("pad_mode", C.c_int),
                ("power", C.c_int), ("log_kind", C.c_int), ("amin", C.c_float), ("ref", C.c_float), ("out_layout", C.c_int)]


class maa_plms_args(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("K_step", C.c_int), ("interval", C.c_int), ("timesteps", C.c_int),
                ("d_cond", C.c_void_p), ("h_alphas_cumprod", C.POINTER(C.c_float)), ("use_graph", C.c_int)]


EXPORTS = [
    "maa_last_error", "maa_version", "maa_ctx_create", "maa_ctx_destroy", "maa_ctx_synchronize",
    "maa_ctx_set_stream", "maa_ctx_set_precision", "maa_ctx_set_cfg_split", "maa_ctx_set_concurrency", "maa_ctx_reload_tuning", "maa_ctx_workspace_bytes", "maa_prof_begin", "maa_prof_end", "maa_unet_create", "maa_unet_destroy",
    "maa_unet_set_context", "maa_unet_forward", "maa_ddim_update", "maa_ddim_sample", "maa_vae_create",
    "maa_vae_destroy", "maa_vae_decode", "maa_vae_decode_spec", "maa_vae_encode_moments", "maa_vocoder_create", "maa_vocoder_destroy",
    "maa_vocoder_forward", "maa_vocoder_forward_f0", "maa_diffnet_create", "maa_diffnet_destroy", "maa_diffnet_forward",
    "maa_plms_sample", "maa_encoder_create", "maa_encoder_destroy", "maa_encoder_text", "maa_encoder_image",
    "maa_encoder_text_cls", "maa_clap_audio_create", "maa_clap_audio_destroy", "maa_clap_audio_embed", "maa_clap_similarity",
    "maa_spectral_create", "maa_spectral_destroy", "maa_spectral_forward", "maa_resampler_create", "maa_resampler_destroy",
    "maa_resampler_forward", "maa_op_linear", "maa_op_conv", "maa_op_groupnorm", "maa_op_layernorm",
    "maa_op_attention", "maa_op_conv_transpose1d", "maa_op_snake_aa", "maa_op_bench_conv", "maa_calib",
]

_lib = None


def load():
    """Load the shared object (once).  Raises MaaError if it is missing: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MaaError("%s not found -- run `python -m audiogpt_amd.build` (hipcc, gfx950)" % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise MaaError("cannot load %s: %s" % (LIB_PATH, e))
    vp, ci, cf, fp = C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_float)
    lib.maa_last_error.restype = C.c_char_p
    lib.maa_version.restype = C.c_char_p
    sig = {
        "maa_ctx_create": [ci, vp, C.POINTER(vp)],
        "maa_ctx_destroy": [vp],
        "maa_ctx_synchronize": [vp],
        "maa_ctx_set_stream": [vp, vp],
        "maa_ctx_workspace_bytes": [vp, C.POINTER(C.c_size_t)],
        "maa_ctx_set_precision": [vp, ci],
        "maa_ctx_set_cfg_split": [vp, ci],
        "maa_ctx_set_concurrency": [vp, ci],
        "maa_ctx_reload_tuning": [vp],
        "maa_prof_begin": [vp, ci],
        "maa_prof_end": [vp, C.POINTER(maa_prof_row), ci, C.POINTER(ci)],
        "maa_unet_create": [vp, C.POINTER(maa_unet_config), C.POINTER(maa_tensor), ci, C.POINTER(vp)],
        "maa_unet_destroy": [vp],
        "maa_unet_set_context": [vp, vp, vp, ci, ci],
        "maa_unet_forward": [vp, vp, vp, vp, ci, ci, ci, vp],
        "maa_ddim_update": [vp, vp, vp, vp, cf, vp, C.c_int64, vp, vp],
        "maa_ddim_sample": [vp, vp, C.POINTER(maa_ddim_args), vp],
        "maa_vae_create": [vp, C.POINTER(maa_vae_config), C.POINTER(maa_tensor), ci, C.POINTER(vp)],
        "maa_vae_destroy": [vp],
        "maa_vae_decode": [vp, vp, vp, ci, ci, ci, cf, vp],
        "maa_vae_decode_spec": [vp, vp, vp, ci, ci, ci, cf, vp],
        "maa_vae_encode_moments": [vp, vp, vp, ci, ci, ci, vp],
        "maa_vocoder_create": [vp, C.POINTER(maa_vocoder_config), C.POINTER(maa_tensor), ci, C.POINTER(vp)],
        "maa_vocoder_destroy": [vp],
        "maa_vocoder_forward": [vp, vp, vp, ci, ci, vp],
        "maa_vocoder_forward_f0": [vp, vp, vp, vp, vp, vp, ci, ci, vp],
        "maa_diffnet_create": [vp, C.POINTER(maa_diffnet_config), C.POINTER(maa_tensor), ci, C.POINTER(vp)],
        "maa_diffnet_destroy": [vp],
        "maa_diffnet_forward": [vp, vp, vp, vp, vp, ci, ci, vp],
        "maa_plms_sample": [vp, vp, C.POINTER(maa_plms_args), vp],
        "maa_encoder_create": [vp, C.POINTER(maa_encoder_config), C.POINTER(maa_tensor), ci, C.POINTER(vp)],
        "maa_encoder_destroy": [vp],
        "maa_encoder_text": [vp, vp, vp, ci, ci, vp],
        "maa_encoder_image": [vp, vp, vp, ci, vp],
        "maa_encoder_text_cls": [vp, vp, vp, ci, ci, vp],
        "maa_clap_audio_create": [vp, C.POINTER(maa_clap_audio_config), C.POINTER(maa_tensor), ci, C.POINTER(vp)],
        "maa_clap_audio_destroy": [vp],
        "maa_clap_audio_embed": [vp, vp, vp, ci, ci, vp, vp],
        "maa_clap_similarity": [vp, vp, vp, ci, ci, ci, cf, vp],
        "maa_spectral_create": [vp, C.POINTER(maa_spectral_config), fp, fp, C.POINTER(vp)],
        "maa_spectral_destroy": [vp],
        "maa_spectral_forward": [vp, vp, vp, ci, ci, vp],
        "maa_resampler_create": [vp, ci, ci, ci, ci, fp, C.POINTER(vp)],
        "maa_resampler_destroy": [vp],
        "maa_resampler_forward": [vp, vp, vp, ci, ci, vp],
        "maa_op_linear": [vp, vp, ci, ci, fp, fp, ci, ci, vp],
        "maa_op_conv": [vp, vp, ci, ci, ci, ci, fp, fp, ci, ci, ci, ci, ci, ci, ci, cf, vp, ci, ci],
        "maa_op_groupnorm": [vp, vp, ci, ci, ci, fp, fp, cf, ci, vp],
        "maa_op_layernorm": [vp, vp, ci, ci, fp, fp, cf, vp],
        "maa_op_attention": [vp, vp, vp, vp, ci, ci, ci, ci, ci, cf, vp],
        "maa_op_conv_transpose1d": [vp, vp, ci, ci, ci, fp, fp, ci, ci, ci, cf, vp],
        "maa_op_snake_aa": [vp, vp, ci, ci, ci, fp, fp, ci, vp],
        "maa_op_bench_conv": [vp, ci, ci, ci, ci, ci, ci, ci, ci, C.POINTER(C.c_float)],
        "maa_calib": [vp, ci, C.POINTER(C.c_double)],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ci
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise MaaError("libaudiogpt_mi355x: %s (status %d)" % (load().maa_last_error().decode(), status))


def host_f32(t):
    """A contiguous fp32 CPU tensor and its float* (keep the tensor alive while the pointer is used)."""
    t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
    return t, C.cast(t.data_ptr(), C.POINTER(C.c_float))


def tensor_list(state_dict):
    """state_dict {name: tensor} -> (maa_tensor array, n, keepalive)."""
    items = list(state_dict.items())
    arr = (maa_tensor * len(items))()
    keep = []
    for i, (name, t) in enumerate(items):
        ht, ptr = host_f32(t)
        bname = name.encode()
        keep.append((ht, bname))
        arr[i].name = bname
        arr[i].data = ptr
        arr[i].ndim = ht.dim()
        for d, s in enumerate(ht.shape):
            arr[i].shape[d] = s
    return arr, len(items), keep


def dptr(t):
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor."""
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())
