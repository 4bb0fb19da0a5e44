"""Shape source of truth for the Make-An-Audio hot path, as plain dicts.

The reference keeps these in OmegaConf YAMLs (omegaconf is not available here):
  * T2A      text_to_audio/Make_An_Audio/configs/text_to_audio/txt2audio_args.yaml:1-78
  * I2A      text_to_audio/Make_An_Audio/configs/img_to_audio/img2audio_args.yaml:1-77
  * Inpaint  text_to_audio/Make_An_Audio/configs/inpaint/txt2audio_args.yaml:1-68
  * HiFi-GAN text_to_audio/Make_An_Audio/vocoder/logs/hifi_0127/args.yml:1-47,
             NeuralSeq/egs/egs_bases/tts/vocoder/hifigan.yaml:3-10 (uic 512),
             NeuralSeq/configs/tts/hifigan.yaml:3-10 (uic 128)
"""
from copy import deepcopy

# ---------------------------------------------------------------- UNet variants
UNET_T2A = dict(
    variant="t2a",
    in_channels=4, out_channels=4, model_channels=320,
    attention_resolutions=(1, 2), num_res_blocks=2, channel_mult=(1, 2),
    num_heads=8, num_head_channels=-1,
    use_spatial_transformer=True, transformer_depth=1, context_dim=1024,
    legacy=False, resblock_updown=False,
    add_context_to_emb=False,          # custom_openaimodel.py:352-354 (I2A only)
)

UNET_I2A = dict(
    variant="i2a",
    in_channels=4, out_channels=4, model_channels=256,
    attention_resolutions=(1, 2), num_res_blocks=2, channel_mult=(1, 2),
    num_heads=-1, num_head_channels=32,
    use_spatial_transformer=True, transformer_depth=1, context_dim=1024,
    legacy=True, resblock_updown=False,
    add_context_to_emb=True,
)

UNET_INPAINT = dict(
    variant="inpaint",
    in_channels=9, out_channels=4, model_channels=320,
    attention_resolutions=(1, 2), num_res_blocks=2, channel_mult=(1, 2),
    num_heads=8, num_head_channels=-1,
    use_spatial_transformer=False, transformer_depth=1, context_dim=None,
    legacy=True, resblock_updown=True,
    add_context_to_emb=False,
)

# ---------------------------------------------------------------- VAE (AutoencoderKL ddconfig)
VAE_DDCONFIG = dict(
    double_z=True, z_channels=4, resolution=848, in_channels=1, out_ch=1,
    ch=128, ch_mult=(1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(106, 212),
    embed_dim=4,
)

# ---------------------------------------------------------------- diffusion schedules
LDM_T2A = dict(
    unet=UNET_T2A, vae=VAE_DDCONFIG, timesteps=1000,
    linear_start=0.00085, linear_end=0.0120,
    conditioning_key="crossattn", latent_shape=(4, 10, 78), scale_factor=1.0,
    sample_rate=16000, hop=256,
)
LDM_I2A = dict(
    unet=UNET_I2A, vae=VAE_DDCONFIG, timesteps=1000,
    linear_start=0.00085, linear_end=0.0120,
    conditioning_key="crossattn", latent_shape=(4, 10, 78), scale_factor=1.0,
    sample_rate=16000, hop=256,
)
LDM_INPAINT = dict(
    unet=UNET_INPAINT, vae=VAE_DDCONFIG, timesteps=1000,
    linear_start=0.0015, linear_end=0.0205,
    conditioning_key="concat", latent_shape=(4, 10, 106), scale_factor=1.0,
    sample_rate=16000, hop=256,
)

# ---------------------------------------------------------------- vocoders
HIFIGAN_16K = dict(           # MAA/vocoder/logs/hifi_0127/args.yml
    kind="hifigan", num_mels=80, upsample_initial_channel=512,
    upsample_rates=(8, 8, 2, 2), upsample_kernel_sizes=(16, 16, 4, 4),
    resblock="1", resblock_kernel_sizes=(3, 7, 11),
    resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)),
    sampling_rate=16000,
)
HIFIGAN_NS_512 = dict(HIFIGAN_16K, sampling_rate=22050)                    # egs_bases/tts/vocoder/hifigan.yaml
HIFIGAN_NS_128 = dict(HIFIGAN_16K, sampling_rate=22050, upsample_initial_channel=128)  # configs/tts/hifigan.yaml

# The generators' other residual block (`resblock: "2"`: ResBlock2, hifigan.py:70-91 / modules.py:62-83; AMPBlock2,
# bigvgan/models.py:90-132): no shipped config selects it -- these are HiFi-GAN's published V3 shapes (hop 256), and
# the same with BigVGAN's plain `snake` activation (activations.py:8-59), which BIGVGAN_16K does not exercise either.
HIFIGAN_RB2 = dict(HIFIGAN_16K, upsample_initial_channel=256, upsample_rates=(8, 8, 4), upsample_kernel_sizes=(16, 16, 8),
                   resblock="2", resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)))
BIGVGAN_RB2 = dict(HIFIGAN_RB2, kind="bigvgan", activation="snake", snake_logscale=False)

# NSF (f0-conditioned) generator of the singing tools: checkpoints/0109_hifigan_bigpopcs_hop128 is not shipped, so the
# rates are the hop-128 / 24 kHz factorisation assumed here (egs_bases/svs/popcs_ds_beta6.yaml:6-7,54-55,70); used by the
# oracle groundwork for SURVEY 8f/N1 only.
HIFIGAN_NSF_24K = dict(HIFIGAN_16K, sampling_rate=24000, upsample_rates=(8, 4, 2, 2), upsample_kernel_sizes=(16, 8, 4, 4),
                       use_pitch_embed=True)

# DiffSinger denoiser + PLMS loop of the T2S tool (egs_bases/svs/base.yaml:3-5, midi/e2e/opencpop/ds1000.yaml:22-36;
# checkpoints/0831_opencpop_ds1000): oracle groundwork for SURVEY 8f/N2 only.
DIFFSINGER_DS1000 = dict(in_dims=80, hidden_size=256, residual_layers=20, residual_channels=256, dilation_cycle_length=4,
                         timesteps=1000, K_step=1000, max_beta=0.02, pndm_speedup=10)

# BigVGAN's args.yml (vocoder/logs/bigv16k53w) does not ship with the reference
# (SURVEY.md section 0.3); these are the generator defaults it is exercised with here.
BIGVGAN_16K = dict(
    HIFIGAN_16K, kind="bigvgan", activation="snakebeta", snake_logscale=True,
)

# ---------------------------------------------------------------- conditioning encoders
# CLAP text branch: transformers' bert-base-uncased + Projection 768 -> 1024 (encoders/CLAP/config.yml, CLAP/clap.py:8-45);
# FrozenCLAPEmbedder pads / truncates to 77 tokens (encoders/modules.py:175,204-206)
CLAP_TEXT = dict(kind="text", layers=12, width=768, heads=12, mlp_dim=3072, d_proj=1024, vocab=30522, max_positions=512,
                 type_vocab=2, ln_eps=1e-12, max_length=77)
# OpenCLIP ViT-H-14 image tower (open_clip model config "ViT-H-14": width 1280, 32 layers, head width 80, mlp ratio 4,
# patch 14, image 224, embed_dim 1024), as FrozenGlobalNormOpenCLIPEmbedder builds it (encoders/modules.py:319-321)
OPENCLIP_VITH14_IMAGE = dict(kind="image", layers=32, width=1280, heads=16, mlp_dim=5120, d_proj=1024, patch=14, image=224,
                             ln_eps=1e-5)
# OpenCLIP ViT-H-14 text tower (open_clip model config "ViT-H-14": context 77, vocab 49408, width 1024, 16 heads,
# 24 layers, embed_dim 1024): FrozenGlobalNormOpenCLIPEmbedder.forward -- I2A's unconditional prompt (audio-chatgpt.py:238)
OPENCLIP_VITH14_TEXT = dict(kind="clip_text", layers=24, width=1024, heads=16, mlp_dim=4096, d_proj=1024, vocab=49408,
                            max_positions=77, ln_eps=1e-5, sot=49406, eot=49407)
# CLAP audio branch of the best-of-n scorer (wav_evaluation/models/CLAPWrapper.py, useful_ckpts/CLAP/config.yml): Cnn14 on
# a 64-bin log-mel at 44.1 kHz (window 1024, hop 320), 2048-d embedding -> Projection -> 1024.  `frames` is only the
# length of the golden test case (690 frames = 5 s); the scorer's own clips are `duration` * the INPUT sample rate long.
CLAP_AUDIO_CNN14 = dict(mel_bins=64, channels=(64, 128, 256, 512, 1024, 2048), out_emb=2048, d_proj=1024, classes_num=527,
                        sample_rate=44100, window_size=1024, hop_size=320, fmin=50, fmax=14000, frames=690)


# The whole scorer as T2A.select_best_audio builds it (audio-chatgpt.py:185-199; useful_ckpts/CLAP/config.yml):
# text side = the same BERT + Projection architecture as CLAP_TEXT, padded to text_len with an attention mask
CLAP_SCORER = dict(text=dict(CLAP_TEXT, max_length=100), audio=CLAP_AUDIO_CNN14, sampling_rate=44100, duration=9, text_len=100,
                   window_size=1024, hop_size=320, mel_bins=64, fmin=50, fmax=14000, amin=1e-10, ref=1.0,
                   resample=dict(lowpass_filter_width=6, rolloff=0.99))


def small(cfg, **over):
    """A reduced copy of a config for quick tests."""
    c = deepcopy(cfg)
    c.update(over)
    return c


def hop_size(voc_cfg):
    h = 1
    for u in voc_cfg["upsample_rates"]:
        h *= u
    return h
