/* libaudiogpt_mi355x -- C ABI of the MI355X-native Make-An-Audio generation backend.
 *
 * The reference (AIGC-Audio/AudioGPT) is pure Python/PyTorch and has NO FFI / plugin layer for this path
 * (SURVEY.md 8b): what this header replaces are the Python methods the tool classes call.  Each entry
 * point names the reference interface it stands in for; `INTEGRATION.md` shows the ctypes binding a
 * maintainer adds on the reference side (audiogpt_amd/_lib.py is that binding, shipped).
 *
 * Conventions
 *   - every function returns 0 on success, a negative maa_status otherwise; maa_last_error() returns a
 *     thread-local message.  No C++ exception crosses the boundary.
 *   - tensors are plain pointers + sizes.  Device pointers are fp32 and live on the context's device;
 *     the caller owns every buffer, the library owns only the opaque handles.
 *   - tensor layouts at the boundary are the reference's: images NCHW, mels [B, n_mels, T], waves [B, T*hop],
 *     cross-attention context [B, L, context_dim].  (Inside, activations are channels-last.)
 *   - all launches are asynchronous on the context's HIP stream; only maa_ctx_synchronize blocks.
 *   - one context per (device, stream); calls on one context must be serialised by the caller
 *     (the reference's DDIMSampler is not re-entrant either: ddim.py:27-56 re-registers buffers per call).
 *   - weights are passed as HOST pointers in the reference's state_dict layout (names + shapes + fp32 data);
 *     the library repacks them for its kernels and uploads them.
 */
#ifndef MAA_H_
#define MAA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum maa_status {
    MAA_OK = 0,
    MAA_ERR_INVALID = -1,   /* bad argument / shape / missing weight */
    MAA_ERR_HIP = -2,       /* HIP runtime failure */
    MAA_ERR_INTERNAL = -3
} maa_status;

typedef struct maa_ctx maa_ctx;         /* device + stream + workspace */
typedef struct maa_unet maa_unet;       /* UNetModel weights + plan */
typedef struct maa_vae maa_vae;         /* AutoencoderKL decoder (+ encoder) */
typedef struct maa_vocoder maa_vocoder; /* HiFi-GAN / BigVGAN generator */

/* thread-local message of the last failing call on this thread */
const char* maa_last_error(void);
/* "libaudiogpt_mi355x <version> gfx950" */
const char* maa_version(void);

/* ---- context ------------------------------------------------------------------------------------
 * stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the default stream. */
int maa_ctx_create(int device_id, void* hip_stream, maa_ctx** out);
int maa_ctx_destroy(maa_ctx* ctx);
int maa_ctx_synchronize(maa_ctx* ctx);
int maa_ctx_set_stream(maa_ctx* ctx, void* hip_stream);
/* Arithmetic of the contractions for models CREATED after this call (it fixes their packed weight layout) and
 * for the maa_op_* entry points:
 *   0  exact fp32: v_mfma_f32_32x32x2_f32, bit-for-bit a k-ordered fmaf chain (default)
 *   1  bf16x3: fp32 operands split on the fly into bf16 hi + lo, lo*hi + hi*lo + hi*hi on the bf16 MFMA with fp32
 *      accumulation (~2^-16 per product; meets the fp32 parity gates at 5.3x the fp32 MFMA rate)
 *   2  bf16: operands rounded to bf16, fp32 accumulation (throughput mode, error reported not gated)
 * Storage, normalisations, softmax and every epilogue stay fp32 in all modes. */
int maa_ctx_set_precision(maa_ctx* ctx, int mode);
/* Classifier-free guidance inside maa_ddim_sample (ddim.py:177-199 evaluates the model on cat([x] * 2) in one call): the
 * unconditional and the conditional half of that batch are independent trajectories until the combine, and the library runs
 * them as two lanes -- two branches of the captured step graph, each half on its own stream and workspace -- because one batch
 * of 8 prompts leaves much of the chip idle.  Bit-identical to the one-stream form (every kernel is batch-invariant).
 *   1 two lanes, 0 one stream, -1 the default policy: two lanes unless the caller has said that the chip is kept full from
 * outside (maa_ctx_set_concurrency(ctx, n >= 3) below; measured +4 % for one batch owning the GPU, +3.8 % with two contexts in
 * flight, -24 % with three, profiles/r5/r5_call1_cfg_lanes_ab.txt). */
int maa_ctx_set_cfg_split(maa_ctx* ctx, int mode);
/* The serving arrangement, as a hint (no counterpart in the reference, which runs one request at a time): n = how many contexts'
 * launches the caller keeps in flight on this device.  n >= 3: the chip is kept full from outside, so a launch costs the sum of
 * its workgroups' time rather than the rounds its own grid makes -- a guided DDIM step runs on one stream (as cfg_split 0: six
 * concurrent lanes lose 24 %) and the short-K contractions take the tile with the least total workgroup time (128 x 64 / 128 x 128
 * where one launch alone would take 64 x 64: +0.9 % with three batches in flight, -7 % for a batch alone,
 * profiles/r6_call6_tile_mode_ab.txt).  n = 1 or 2: this context (nearly) owns the GPU -- two CFG lanes (+4 % / +3.8 %), tiles by
 * least launch time.  -1 (default) = not told, treated as 1: a server that keeps three or more batches in flight on as many
 * contexts SHOULD call this (or maa_ctx_set_cfg_split(ctx, 0)) -- left at the default it runs six CFG lanes and loses 24 %.  Every
 * choice is bit-identical; a kept DDIM step graph is dropped when the arrangement changes.  maa_ctx_set_cfg_split overrides the
 * lanes part. */
int maa_ctx_set_concurrency(maa_ctx* ctx, int n);
/* The test / A-B knobs of the environment (MAA_PP, MAA_PP1, MAA_PP_S, MAA_PP_TILE_MAJOR, MAA_UP2, MAA_DMA2, MAA_NO_DMA, MAA_HALO,
 * MAA_OP_PRESPLIT, MAA_GN_TWO_PASS) are parsed in one place, when a context is created; this parses them again
 * (and drops the step graph the sampler keeps).  A malformed MAA_DMA2 value fails here (and in maa_ctx_create) with a message
 * naming the variable.  For tests and A/B runs; a -DMAA_NO_TUNING build ignores the environment. */
int maa_ctx_reload_tuning(maa_ctx* ctx);
/* bytes currently reserved for the activation workspace, the second CFG lane's arena included (it exists once a guided sample()
 * has run with two lanes and is kept for the context's lifetime) */
int maa_ctx_workspace_bytes(maa_ctx* ctx, size_t* out);

/* ---- per-kernel timing (replaces the reference's utils.Timer, NeuralSeq/utils/__init__.py:222-237) ----
 * Between begin and end every kernel launch on this context is bracketed by a hipEvent pair on the context's
 * stream; end synchronises and returns one row per kernel (aggregated), with the ALGORITHMIC flops / bytes of
 * the launches.  Launches captured into a hipGraph are not timed (run with use_graph = 0 while profiling). */
typedef struct maa_prof_row {
    char name[48];
    int64_t launches;
    double ms;       /* sum of launch durations */
    double flops;    /* sum of algorithmic flops (2*M*N*K for the contractions) */
    double bytes;    /* sum of algorithmic bytes (compulsory reads + writes) */
} maa_prof_row;
int maa_prof_begin(maa_ctx* ctx, int detail);   /* detail 1: contraction rows are keyed by problem shape */
int maa_prof_end(maa_ctx* ctx, maa_prof_row* rows, int max_rows, int* n_rows);

/* one named fp32 host tensor of a reference state_dict */
typedef struct maa_tensor {
    const char* name;        /* e.g. "input_blocks.1.0.in_layers.2.weight" */
    const float* data;       /* host pointer, contiguous */
    int ndim;
    int64_t shape[6];
} maa_tensor;

/* ---- UNet ---------------------------------------------------------------------------------------
 * Mirrors the constructor arguments of ldm.modules.diffusionmodules.openaimodel.UNetModel
 * (openaimodel.py:443-470) as used by the three shipped configs (txt2audio_args.yaml:31-50,
 * img2audio_args.yaml:31-49, inpaint/txt2audio_args.yaml:30-45). */
typedef struct maa_unet_config {
    int in_channels, out_channels, model_channels;
    int num_res_blocks;
    int n_channel_mult, channel_mult[8];
    int n_attention_resolutions, attention_resolutions[8];
    int num_heads, num_head_channels;          /* -1 = unset, as in the reference */
    int use_spatial_transformer, transformer_depth, context_dim;
    int legacy, resblock_updown;
    int add_context_to_emb;                    /* custom_openaimodel.py:352-354 (I2A) */
} maa_unet_config;

/* replaces: instantiate_from_config(unet_config) + load_state_dict (audio-chatgpt.py:147-152);
 * tensors: the `model.diffusion_model.`-relative entries of the checkpoint */
int maa_unet_create(maa_ctx* ctx, const maa_unet_config* cfg, const maa_tensor* tensors, int n_tensors,
                    maa_unet** out);
int maa_unet_destroy(maa_unet* u);
/* Cross-attention context for the following forwards: d_context [B, L, context_dim] (device).  The K/V
 * projections of every transformer block are computed here, once, and reused by each forward -- the
 * context is constant over a DDIM trajectory (ddim.py:177-198 rebuilds c_in from the same c/uc every step). */
int maa_unet_set_context(maa_ctx* ctx, maa_unet* u, const float* d_context, int B, int L);
/* replaces: UNetModel.forward(x, timesteps, context) (openaimodel.py:711-744 / custom_openaimodel.py:331-368)
 * d_x [B, Cin, H, W], d_t [B] (timesteps as fp32), d_out [B, Cout, H, W].  For the I2A variant the
 * context passed to set_context is also added to the time embedding. */
int maa_unet_forward(maa_ctx* ctx, maa_unet* u, const float* d_x, const float* d_t, int B, int H, int W,
                     float* d_out);

/* ---- DDIM ---------------------------------------------------------------------------------------
 * replaces: DDIMSampler.p_sample_ddim's elementwise tail (ddim.py:199, 210-225), eta = 0 or with caller noise:
 *   e = eu + scale*(ec - eu) (ec may be NULL: no guidance);  x0 = (x - sqrt(1-a_t) e)/sqrt(a_t);
 *   x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma^2) e
 * d_coef: device float[4] = {a_t, a_prev, sigma_t, sqrt(1-a_t)} exactly as the sampler's fp32 tables hold them. */
int maa_ddim_update(maa_ctx* ctx, const float* d_x, const float* d_eps_uncond, const float* d_eps_cond, float scale,
                    const float* d_coef, int64_t n, float* d_x_prev, float* d_pred_x0);

/* replaces: DDIMSampler.ddim_sampling (ddim.py:118-166), including mask / x0 blending, eta > 0 with the caller's noise and
 * the logged intermediates (score correctors, quantisation, dropout noise and host callbacks need host code between steps:
 * the Python sampler runs such calls step by step over maa_unet_forward + maa_ddim_update):
 * runs S steps on the device without host round trips.
 *   d_x [B, C, H, W] in/out latent (x_T in, x_0 out)
 *   d_cond / d_uncond [B, L, context_dim] (crossattn; d_uncond NULL or scale == 1 -> no CFG), or for the
 *   concat-conditioned inpaint model d_concat [B, Cc, H, W] (cat([x, c]) -> UNet, ddpm.py:1404-1406)
 *   h_timesteps [S] (ascending DDIM timesteps), h_alphas / h_alphas_prev [S] fp32 tables (host) */
typedef struct maa_ddim_args {
    int S, B, C, H, W;
    float scale;
    const float* d_cond;
    const float* d_uncond;
    int L;
    const float* d_concat;
    int Cc;
    const int32_t* h_timesteps;
    const float* h_alphas;
    const float* h_alphas_prev;
    int use_graph;          /* capture one step into a hipGraph and replay it */
    /* ---- the rest of DDIMSampler.sample's signature (ddim.py:59-115); all optional (NULL / 0) ----
     * mask / x0 (ddim.py:147-150): before every step img = q_sample(x0, t) * mask + (1 - mask) * img, with
     *   q_sample = h_sqrt_ac[i] x0 + h_sqrt_1mac[i] z (ddpm.py:272-275); d_mask, d_x0 [B, C, H, W]; d_noise_q [S][B, C, H, W] =
     *   the S draws of randn_like(x0) in the order the loop makes them (first step first) */
    const float* d_mask;
    const float* d_x0;
    const float* d_noise_q;
    const float* h_sqrt_ac;      /* [S] sqrt(alphas_cumprod[t_i]), sqrt(1 - alphas_cumprod[t_i]) per DDIM index, fp32 as the */
    const float* h_sqrt_1mac;    /*     model's buffers hold them (ddpm.py:139-140) */
    /* eta > 0 (ddim.py:210-225): x_prev += sigma_t * z * temperature; h_sigmas [S] per DDIM index as make_schedule forms them,
     *   d_noise_p [S][B, C, H, W] = the S draws of noise_like(x.shape) in loop order */
    const float* h_sigmas;
    const float* d_noise_p;
    float temperature;
    /* intermediates (ddim.py:158-163): after the step of DDIM index i with i % log_every_t == 0 or i == S - 1 the new latent and
     *   pred_x0 are copied to d_log_x / d_log_x0 [n_log][B, C, H, W], in loop order; n_log must equal the number of such steps */
    int log_every_t;
    int n_log;
    float* d_log_x;
    float* d_log_x0;
} maa_ddim_args;
int maa_ddim_sample(maa_ctx* ctx, maa_unet* u, const maa_ddim_args* args, float* d_x);

/* ---- VAE ----------------------------------------------------------------------------------------
 * ddconfig of ldm.models.autoencoder.AutoencoderKL (txt2audio_args.yaml:54-68) */
typedef struct maa_vae_config {
    int ch, out_ch, in_channels, z_channels, embed_dim, resolution, num_res_blocks, double_z;
    int n_ch_mult, ch_mult[8];
    int n_attn_resolutions, attn_resolutions[8];
} maa_vae_config;
/* tensors: the `first_stage_model.`-relative entries (decoder.*, post_quant_conv.*; encoder.*, quant_conv.* optional) */
int maa_vae_create(maa_ctx* ctx, const maa_vae_config* cfg, const maa_tensor* tensors, int n_tensors, maa_vae** out);
int maa_vae_destroy(maa_vae* v);
/* replaces: LatentDiffusion_audio.decode_first_stage (ddpm_audio.py:352-359) -> AutoencoderKL.decode
 * (autoencoder.py:351-354): d_z [B, 4, h, w] -> d_mel [B, out_ch, 8h, 8w]; inv_scale = 1/scale_factor */
int maa_vae_decode(maa_ctx* ctx, maa_vae* v, const float* d_z, int B, int h, int w, float inv_scale, float* d_mel);
/* replaces: decode_first_stage followed by the tools' `torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)` (audio-chatgpt.py:175-176,
 * 254-255, 521-522): d_z [B, 4, h, w] -> d_spec [B, 8h, 8w] in [0, 1], what the vocoders take (one-channel decoders only) */
int maa_vae_decode_spec(maa_ctx* ctx, maa_vae* v, const float* d_z, int B, int h, int w, float inv_scale, float* d_spec);
/* replaces: AutoencoderKL.encode's moments (autoencoder.py:345-349): d_mel [B, 1, H, W] ->
 * d_moments [B, 2*embed_dim, H/8, W/8] = (mean | logvar, unclamped) */
int maa_vae_encode_moments(maa_ctx* ctx, maa_vae* v, const float* d_mel, int B, int H, int W, float* d_moments);

/* ---- vocoder ------------------------------------------------------------------------------------ */
typedef struct maa_vocoder_config {
    int kind;                      /* 0 HiFi-GAN (leaky-ReLU MRF), 1 BigVGAN (anti-aliased snake MRF) */
    int num_mels, upsample_initial_channel;
    int n_upsamples, upsample_rates[8], upsample_kernel_sizes[8];
    int n_kernels, resblock_kernel_sizes[8];
    int n_dilations, resblock_dilation_sizes[8][8];
    int snake_beta, snake_logscale; /* BigVGAN: activation == "snakebeta", snake_logscale */
    /* NSF branch (h['use_pitch_embed'], NeuralSeq/modules/hifigan/hifigan.py:111-115,124-132): harmonic-plus-noise source
     * of `harmonic_num` overtones at `sampling_rate` (h['audio_sample_rate']), added through noise_convs after each ups[i] */
    int use_pitch_embed, sampling_rate, harmonic_num;
    /* h.resblock: 1 (or 0) = ResBlock1 / AMPBlock1 (`convs1` + `convs2`), 2 = ResBlock2 / AMPBlock2 (`convs`: one dilated
     * convolution per residual step; hifigan.py:70-91,119, vocoder/hifigan/modules.py:62-83,93, bigvgan/models.py:90-132,146) */
    int resblock;
} maa_vocoder_config;
/* tensors: generator state_dict (weight_g/weight_v pairs or folded `weight`), keys as
 * NeuralSeq/modules/hifigan/hifigan.py:104-142 / vocoder/bigvgan/models.py:133-179 */
int maa_vocoder_create(maa_ctx* ctx, const maa_vocoder_config* cfg, const maa_tensor* tensors, int n_tensors,
                       maa_vocoder** out);
int maa_vocoder_destroy(maa_vocoder* v);
/* replaces: HifiGanGenerator.forward(x, f0=None) (hifigan.py:144-169), Generator.forward
 * (vocoder/hifigan/modules.py:111-127), BigVGAN.forward (bigvgan/models.py:181-203):
 * d_mel [B, num_mels, T] -> d_wav [B, T*hop] */
int maa_vocoder_forward(maa_ctx* ctx, maa_vocoder* v, const float* d_mel, int B, int T, float* d_wav);
/* replaces: HifiGanGenerator.forward(x, f0) with use_pitch_embed (hifigan.py:144-169: f0_upsamp, m_source, noise_convs)
 * and SineGen / SourceModuleHnNSF (NeuralSeq/modules/parallel_wavegan/models/source.py:399-436, 526-535).
 * d_f0 [B, T] in Hz (0 = unvoiced).  The two random tensors SineGen.forward draws are inputs:
 * d_rand_ini [B, harmonic_num + 1] ~ U[0,1) (initial phase of the overtones; column 0 is ignored) and
 * d_noise [B, T*hop, harmonic_num + 1] ~ N(0,1) (additive noise). */
int maa_vocoder_forward_f0(maa_ctx* ctx, maa_vocoder* v, const float* d_mel, const float* d_f0,
                           const float* d_rand_ini, const float* d_noise, int B, int T, float* d_wav);

/* ---- DiffSinger denoiser + PLMS loop (the singing-voice tool's diffusion hot loop) ---------------
 * DiffNet: NeuralSeq/modules/diff/net.py:84-130 (hparams hidden_size / residual_layers / residual_channels /
 * dilation_cycle_length, in_dims = audio_num_mel_bins) */
typedef struct maa_diffnet_config {
    int in_dims, hidden_size, residual_layers, residual_channels, dilation_cycle_length;
} maa_diffnet_config;
typedef struct maa_diffnet maa_diffnet;
/* tensors: the `denoise_fn.`-relative state_dict of GaussianDiffusion (input_projection.*, mlp.{0,2}.*,
 * residual_layers.{i}.{dilated_conv,diffusion_projection,conditioner_projection,output_projection}.*, skip_projection.*,
 * output_projection.*) */
int maa_diffnet_create(maa_ctx* ctx, const maa_diffnet_config* cfg, const maa_tensor* tensors, int n_tensors,
                       maa_diffnet** out);
int maa_diffnet_destroy(maa_diffnet* d);
/* replaces: DiffNet.forward(spec, diffusion_step, cond) (net.py:107-130):
 * d_spec [B, 1, in_dims, T], d_t [B] (the integer diffusion step as float), d_cond [B, hidden_size, T] -> d_eps [B, 1, in_dims, T] */
int maa_diffnet_forward(maa_ctx* ctx, maa_diffnet* d, const float* d_spec, const float* d_t, const float* d_cond, int B,
                        int T, float* d_eps);
/* replaces: the pndm_speedup branch of GaussianDiffusion.forward(infer=True) (shallow_diffusion_tts.py:262-269) with
 * p_sample_plms (:166-201): t = K_step - interval, ..., 0; d_x [B, 1, in_dims, T] holds x_K on entry and x_0 on return.
 * h_alphas_cumprod [timesteps]: the fp32 buffer of the reference (:82-96). */
typedef struct maa_plms_args {
    int B, T, K_step, interval, timesteps;
    const float* d_cond;              /* [B, hidden_size, T] */
    const float* h_alphas_cumprod;    /* host, [timesteps] */
    int use_graph;
} maa_plms_args;
int maa_plms_sample(maa_ctx* ctx, maa_diffnet* d, const maa_plms_args* args, float* d_x);

/* ---- conditioning encoders (the step before the sampler: text / image -> cross-attention context) ----------
 * kind 0: the CLAP text branch as FrozenCLAPEmbedder.encode runs it (ldm/modules/encoders/modules.py:204-211):
 *   transformers BertModel (bert-base-uncased, built by TextEncoder, CLAP/clap.py:41-45) on input_ids alone -- no
 *   attention mask, token type 0 -- then Projection (CLAP/clap.py:8-20) on EVERY token: [B, L] ids -> [B, L, d_proj].
 *   tensors: `caption_encoder.`-relative keys: base.embeddings.{word,position,token_type}_embeddings.weight,
 *   base.embeddings.LayerNorm.*, base.encoder.layer.{i}.attention.self.{query,key,value}.*,
 *   .attention.output.{dense,LayerNorm}.*, .intermediate.dense.*, .output.{dense,LayerNorm}.*,
 *   projection.{linear1,linear2}.weight, projection.layer_norm.*
 * kind 1: the OpenCLIP image tower behind FrozenGlobalNormOpenCLIPEmbedder.forward_img (modules.py:340-343):
 *   open_clip VisionTransformer (ViT-H-14: patch 14, width 1280, 32 layers, 16 heads, mlp 5120, GELU), CLS token ->
 *   ln_post -> @ proj, then z / ||z||: [B, 3, image, image] (already preprocessed) -> [B, d_proj].
 *   tensors: `model.visual.`-relative open_clip keys: conv1.weight, class_embedding, positional_embedding, ln_pre.*,
 *   transformer.resblocks.{i}.{ln_1,ln_2}.*, .attn.{in_proj_weight,in_proj_bias}, .attn.out_proj.*, .mlp.{c_fc,c_proj}.*,
 *   ln_post.*, proj ([width, d_proj])
 * kind 2: the OpenCLIP text tower behind FrozenGlobalNormOpenCLIPEmbedder.forward (modules.py:334-338; the image-to-audio
 *   tool encodes its unconditional prompt "" with it, audio-chatgpt.py:238): open_clip CLIP.encode_text (token +
 *   positional embedding, pre-LayerNorm blocks under a causal mask, ln_final, features at the end-of-text token = argmax
 *   of the ids, @ text_projection), then z / ||z||: [B, L] ids -> [B, d_proj].
 *   tensors: `model.`-relative open_clip keys: token_embedding.weight, positional_embedding, transformer.resblocks.{i}.*
 *   (as kind 1), ln_final.*, text_projection ([width, d_proj]) */
typedef struct maa_encoder_config {
    int kind;                       /* 0 = BERT text + CLAP projection, 1 = OpenCLIP ViT image tower, 2 = OpenCLIP text tower */
    int layers, width, heads, mlp_dim, d_proj;
    int vocab, max_positions;       /* kinds 0 and 2 */
    int patch, image;               /* kind 1 */
    float ln_eps;                   /* 1e-12 (BERT) / 1e-5 (ViT) */
} maa_encoder_config;
typedef struct maa_encoder maa_encoder;
int maa_encoder_create(maa_ctx* ctx, const maa_encoder_config* cfg, const maa_tensor* tensors, int n_tensors,
                       maa_encoder** out);
int maa_encoder_destroy(maa_encoder* e);
/* d_ids [B, L] int32 token ids on the device (tokenisation stays on the host);
 * kind 0 -> d_out [B, L, d_proj];  kind 2 -> d_out [B, d_proj], rows L2-normalised */
int maa_encoder_text(maa_ctx* ctx, maa_encoder* e, const int* d_ids, int B, int L, float* d_out);
/* kind 1: d_img [B, 3, image, image] -> d_out [B, d_proj], rows L2-normalised */
int maa_encoder_image(maa_ctx* ctx, maa_encoder* e, const float* d_img, int B, float* d_out);

/* kind 0 only: the scorer's text side -- TextEncoder.forward (wav_evaluation/models/clap.py:49-53: BERT, the [CLS] row,
 * Projection) followed by CLAPWrapper.get_text_embeddings' normalisation (CLAPWrapper.py:177-182).  d_ids holds the
 * UNPADDED token ids: the reference pads to text_len and passes the attention mask, under which the padded keys weigh
 * exactly zero, so the [CLS] row is the same.  -> d_out [B, d_proj], rows L2-normalised */
int maa_encoder_text_cls(maa_ctx* ctx, maa_encoder* e, const int* d_ids, int B, int L, float* d_out);

/* ---- CLAP best-of-n scorer, audio side (T2A.select_best_audio, audio-chatgpt.py:185-199) -------------------
 * replaces: AudioEncoder.forward (wav_evaluation/models/clap.py:22-39) from the log-mel on -- Cnn14.forward after its two
 * extractors (wav_evaluation/models/audio.py:150-176, eval mode) + Projection -- and CLAPWrapper.get_audio_embeddings'
 * normalisation (CLAPWrapper.py:184-189).
 * tensors: `audio_encoder.`-relative keys: base.bn0.*, base.conv_block{1..n}.{conv1,conv2}.weight,
 * base.conv_block{i}.{bn1,bn2}.{weight,bias,running_mean,running_var}, base.fc1.*, projection.{linear1,linear2}.weight,
 * projection.layer_norm.* (fc_audioset and the extractors' frozen tables are not used here) */
typedef struct maa_clap_audio_config {
    int mel_bins, n_blocks, channels[8];
    int out_emb, d_proj;
    float bn_eps;                   /* 1e-5 (nn.BatchNorm2d default) */
} maa_clap_audio_config;
typedef struct maa_clap_audio maa_clap_audio;
int maa_clap_audio_create(maa_ctx* ctx, const maa_clap_audio_config* cfg, const maa_tensor* tensors, int n_tensors,
                          maa_clap_audio** out);
int maa_clap_audio_destroy(maa_clap_audio* a);
/* d_logmel [B, 1, T, mel_bins] -> d_z [B, d_proj] (unit length); d_embedding [B, out_emb] = relu(fc1(.)) or NULL */
int maa_clap_audio_embed(maa_ctx* ctx, maa_clap_audio* a, const float* d_logmel, int B, int T, float* d_embedding,
                         float* d_z);
/* replaces: CLAPWrapper.compute_similarity (CLAPWrapper.py:207-215): d_out [Na, Nt] = scale * audio @ text^T
 * (scale = 1 for use_logit_scale = False, exp(logit_scale) otherwise) */
int maa_clap_similarity(maa_ctx* ctx, const float* d_audio, const float* d_text, int Na, int Nt, int D, float scale,
                        float* d_out);

/* ---- log-mel front ends ------------------------------------------------------------------------------------
 * replaces: torchlibrosa Spectrogram + LogmelFilterBank as Cnn14 builds them (wav_evaluation/models/audio.py:123-131)
 * and TRANSFORMS_16000 (ldm/data/extract_mel_spectrogram.py:15-38, 140-150; Inpaint.gen_mel_audio, audio-chatgpt.py:468-491):
 * centre padding (n_fft / 2 on both sides) -> framed DFT -> re^2 + im^2 (power 2) or its root (power 1) -> mel filter
 * bank -> logarithm.  frames = 1 + n / hop.
 *   h_basis [2 n_freq][n_fft]: rows 0..n_freq-1 the real, n_freq.. the imaginary DFT rows, analysis window folded in
 *     (= torchlibrosa's stft.conv_real.weight / conv_imag.weight, which a CLAP checkpoint carries)
 *   h_melw  [n_mels][n_freq]: the mel filter bank (librosa.filters.mel; = logmel_extractor.melW transposed)
 *   log_kind 0: 10 log10(max(amin, x)) - 10 log10(max(amin, ref))     (power_to_db, top_db = None)
 *   log_kind 1: clip((20 log10(max(amin, x)) - 20 + 100) / 100, 0, 1)  (TRANSFORMS_16000)
 *   out_layout 0: d_out [B, frames, n_mels] (= Cnn14's [B, 1, T, mel_bins]);  1: [B, n_mels, frames] (the LDM's mel image)
 * Always exact fp32, whatever the context's precision mode. */
typedef struct maa_spectral_config {
    int n_fft, hop, n_freq, n_mels;
    int pad_mode;                   /* 0 zeros (librosa >= 0.10), 1 reflect (torchlibrosa; librosa <= 0.9.2) */
    int power;                      /* 1 or 2 */
    int log_kind;
    float amin, ref;
    int out_layout;
} maa_spectral_config;
typedef struct maa_spectral maa_spectral;
int maa_spectral_create(maa_ctx* ctx, const maa_spectral_config* cfg, const float* h_basis, const float* h_melw,
                        maa_spectral** out);
int maa_spectral_destroy(maa_spectral* s);
int maa_spectral_forward(maa_ctx* ctx, maa_spectral* s, const float* d_wav, int B, int n, float* d_out);

/* replaces: torchaudio.transforms.Resample(orig, new) as CLAPWrapper.resample_and_duration applies it
 * (CLAPWrapper.py:103-110): orig / new already divided by their gcd; h_kernels [new][klen], klen = 2 width + orig is
 * torchaudio's sinc kernel bank (built on the host: audiogpt_amd/clap.py).  d_wav [B, n] -> d_out [B, ceil(new n / orig)] */
typedef struct maa_resampler maa_resampler;
int maa_resampler_create(maa_ctx* ctx, int orig, int neu, int width, int klen, const float* h_kernels, maa_resampler** out);
int maa_resampler_destroy(maa_resampler* r);
int maa_resampler_forward(maa_ctx* ctx, maa_resampler* r, const float* d_wav, int B, int n, float* d_out);

/* Box calibration for the benchmark's `box.calib` record (no counterpart in the reference): kind 0 = a fixed register-only
 * loop of dense bf16 MFMAs -> TFLOP/s the box sustains, kind 1 = a fixed 256 MiB device copy -> GB/s (read + written), kind 2 =
 * every workgroup re-reading its own 64 KiB (out of L2) -> GB/s, kind 3 = a 128 MiB buffer re-read by the whole grid (past L2,
 * inside the Infinity Cache) -> GB/s.  Timed with HIP events on the context's stream. */
int maa_calib(maa_ctx* ctx, int kind, double* out_value);

/* ---- single-operator entry points (parity tests and profiling of individual kernels) ------------ */
/* y[M,N] = A[M,K] * W^T (+bias) with W given as torch Linear weight [N,K] on the HOST; A, y on device */
int maa_op_linear(maa_ctx* ctx, const float* d_a, int M, int K, const float* h_w, const float* h_bias, int N,
                  int geglu, float* d_y);
/* conv on channels-first tensors: d_x [B,Cin,H,W] (1-D: H = 1), torch weight [Cout,Cin,KH,KW] on the HOST */
int maa_op_conv(maa_ctx* ctx, const float* d_x, int B, int Cin, int H, int W, const float* h_w, const float* h_bias,
                int Cout, int KH, int KW, int stride, int pad, int dil, int upsample2, float leaky_slope,
                float* d_y, int Ho, int Wo);
/* GroupNorm(32 groups)(+SiLU) on d_x [B,C,HW] */
int maa_op_groupnorm(maa_ctx* ctx, const float* d_x, int B, int C, int HW, const float* h_gamma,
                     const float* h_beta, float eps, int silu, float* d_y);
/* LayerNorm over the last dim of d_x [rows, C] */
int maa_op_layernorm(maa_ctx* ctx, const float* d_x, int rows, int C, const float* h_gamma, const float* h_beta,
                     float eps, float* d_y);
/* softmax(alpha * q k^T) v per head; q [B,Nq,heads*dh], k/v [B,Nk,heads*dh] -> y [B,Nq,heads*dh] */
int maa_op_attention(maa_ctx* ctx, const float* d_q, const float* d_k, const float* d_v, int B, int heads, int dh,
                     int Nq, int Nk, float alpha, float* d_y);
/* ConvTranspose1d, d_x [B,Cin,L], torch weight [Cin,Cout,k] on the HOST, padding (k-stride)/2 -> [B,Cout,L*stride] */
int maa_op_conv_transpose1d(maa_ctx* ctx, const float* d_x, int B, int Cin, int L, const float* h_w,
                            const float* h_bias, int Cout, int k, int stride, float leaky_slope, float* d_y);
/* Kernel-only timing of one 3x3 (taps = 9) or 1x1 (taps = 1) convolution [B,H,W,Cin] -> [B,H,W,Cout] in the
 * context's precision mode on synthetic data: `iters` back-to-back launches between two hipEvents.  pre_split = 1
 * feeds the activation as bf16 hi/lo planes (the GroupNorm/LayerNorm output format of the bf16 modes). */
int maa_op_bench_conv(maa_ctx* ctx, int B, int H, int W, int Cin, int Cout, int taps, int pre_split, int iters,
                      float* ms_per_launch);
/* BigVGAN Activation1d (up2 FIR -> snake(beta) -> down2 FIR) on d_x [B,C,L] */
int maa_op_snake_aa(maa_ctx* ctx, const float* d_x, int B, int C, int L, const float* h_alpha, const float* h_beta,
                    int logscale, float* d_y);

#ifdef __cplusplus
}
#endif
#endif /* MAA_H_ */
