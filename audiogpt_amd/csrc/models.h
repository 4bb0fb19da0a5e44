// Model executors behind the C ABI handles.
#pragma once
#include "../../include/maa.h"
#include "blocks.h"

namespace maa {

class UNet {
public:
    // precision: 0 exact fp32 MFMA, 1 bf16x3 split, 2 plain bf16 operands (fixed at creation: it decides the
    // packed weight layout)
    UNet(const maa_unet_config& cfg, const StateDict& sd, int precision);
    ~UNet();
    void set_context(Ctx& ctx, const float* d_context, int B, int L);
    // classifier-free guidance: context rows [uncond (B) ; cond (B)] assembled in a buffer the UNet owns (it stays
    // valid for later forwards, e.g. the I2A time-embedding add), then set_context over 2B rows
    void set_context_cfg(Ctx& ctx, const float* d_uncond, const float* d_cond, int B, int L);
    // emb_row: optional, the ResBlocks' time-embedding row of this step from emb_table() (one row for all samples: every
    // sample of a DDIM step shares t); null: computed from t (and, I2A, the context) as the reference does per forward
    // batch_off >= 0: this call covers samples [batch_off, batch_off + B) of the batch set_context saw (one lane of a CFG
    // step); x / t / out already point at that sample, the context rows and the cross-attention K/V caches are offset inside
    // cfg_share (with emb_row; a guided DDIM step, whose halves are the same tensor before the first cross-attention): 1 = the
    // batch is cat([x] * 2) on one stream -- samples [B/2, B) repeat [0, B/2) -- and those layers run on one half; 2 / 3 = this
    // call is the unconditional / conditional lane (batch_off 0 / B): lane 2 leaves those layers' outputs in its workspace,
    // lane 3 (called after it) starts from them (unet.cpp forward_body)
    void forward(Ctx& ctx, const float* x_nchw, const float* t, const float* context, int B, int H, int W,
                 float* out_nchw, const float* emb_row = nullptr, int batch_off = -1, int cfg_share = 0);
    // Linear(SiLU(time_embed(timestep_embedding(t)))) of every ResBlock for `rows` timesteps at once -> [rows, emb_width()]
    // (openaimodel.py:725-726, 218-224): the sampling loop computes its S rows once instead of per step.  Not for the
    // I2A variant, whose embedding also takes the sample's context (custom_openaimodel.py:352-354).
    void emb_table(Ctx& ctx, const float* d_t, int rows, float* d_out);
    int emb_width() const;
    const maa_unet_config& config() const;
    size_t weight_bytes() const;
    // context pointer remembered by set_context (needed by the I2A time-embedding add)
    const float* context_ptr = nullptr;
    // addresses / sizes of the UNet-owned buffers a captured forward reads (cross-attention K/V caches, CFG context):
    // appended to a graph cache key
    void graph_key(std::vector<unsigned long long>& key) const;

private:
    struct Impl;
    Impl* impl_;
};

class VAE {
public:
    VAE(const maa_vae_config& cfg, const StateDict& sd, int precision);
    ~VAE();
    void decode(Ctx& ctx, const float* z_nchw, int B, int h, int w, float inv_scale, float* mel_nchw);
    // decode + the tools' clamp((mel + 1) / 2, 0, 1) (audio-chatgpt.py:175-176) in the last pass: spec [B, 8h, 8w]
    void decode_spec(Ctx& ctx, const float* z_nchw, int B, int h, int w, float inv_scale, float* spec);
    void encode_moments(Ctx& ctx, const float* mel_nchw, int B, int H, int W, float* moments_nchw);
    const maa_vae_config& config() const;

private:
    struct Impl;
    Impl* impl_;
};

class Vocoder {
public:
    Vocoder(const maa_vocoder_config& cfg, const StateDict& sd, int precision);
    ~Vocoder();
    void forward(Ctx& ctx, const float* mel, int B, int T, float* wav);
    // NSF branch: f0 [B, T], rand_ini [B, harmonics+1], noise [B, T*hop, harmonics+1] (see maa.h)
    void forward_f0(Ctx& ctx, const float* mel, const float* f0, const float* rand_ini, const float* noise, int B, int T,
                    float* wav);
    int hop() const;

private:
    struct Impl;
    Impl* impl_;
};

class DiffNet {
public:
    DiffNet(const maa_diffnet_config& cfg, const StateDict& sd, int precision);
    ~DiffNet();
    void forward(Ctx& ctx, const float* spec, const float* t, const float* cond, int B, int T, float* out);
    void plms_sample(Ctx& ctx, const maa_plms_args& a, float* d_x);
    const maa_diffnet_config& config() const;

private:
    struct Impl;
    Impl* impl_;
};

class Encoder {
public:
    Encoder(const maa_encoder_config& cfg, const StateDict& sd, int precision);
    ~Encoder();
    void text(Ctx& ctx, const int* d_ids, int B, int L, float* d_out);
    void text_cls(Ctx& ctx, const int* d_ids, int B, int L, float* d_out);
    void image(Ctx& ctx, const float* d_img, int B, float* d_out);
    const maa_encoder_config& config() const;

private:
    struct Impl;
    Impl* impl_;
};

// CLAP audio branch of the best-of-n scorer: Cnn14 from the log-mel on + Projection, unit-length rows (clap_audio.cpp)
class ClapAudio {
public:
    ClapAudio(const maa_clap_audio_config& cfg, const StateDict& sd, int precision);
    ~ClapAudio();
    void embed(Ctx& ctx, const float* d_logmel, int B, int T, float* d_embedding, float* d_z);
    const maa_clap_audio_config& config() const;

private:
    struct Impl;
    Impl* impl_;
};

// framed DFT -> |.|^p -> mel filter bank -> log: the 16 kHz front end of the inpainting tool and the 44.1 kHz one of the
// CLAP scorer (clap_audio.cpp); always exact fp32
class Spectral {
public:
    Spectral(const maa_spectral_config& cfg, const float* h_basis, const float* h_melw);
    ~Spectral();
    void forward(Ctx& ctx, const float* d_wav, int B, int n, float* d_out);
    const maa_spectral_config& config() const;

private:
    struct Impl;
    Impl* impl_;
};

// torchaudio-style polyphase sinc resampler as one strided FIR-bank contraction (clap_audio.cpp); always exact fp32
class Resampler {
public:
    Resampler(int orig, int neu, int width, int klen, const float* h_kernels);
    ~Resampler();
    long long out_length(long long n) const;
    void forward(Ctx& ctx, const float* d_wav, int B, int n, float* d_out);

private:
    struct Impl;
    Impl* impl_;
};

void ddim_sample(Ctx& ctx, UNet& unet, const maa_ddim_args& a, float* d_x);

}  // namespace maa
