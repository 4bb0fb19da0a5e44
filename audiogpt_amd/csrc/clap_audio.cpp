// The device half of SURVEY 8f / N4: the audio branch of the CLAP best-of-n scorer and the two log-mel front ends.
//
//   ClapAudio   wav_evaluation/models/audio.py:107-180 (Cnn14.forward after its two extractors: bn0 over the mel bins, six
//               ConvBlocks = 3x3 conv without bias -> BatchNorm -> ReLU, twice, then 2x2 average pooling (none after the
//               last), mean over frequency, max + mean over time, relu(fc1)), clap.py:8-20 (Projection),
//               CLAPWrapper.py:186-191 (unit length).  Eval-mode BatchNorm is folded into the convolution on the host:
//               w' = w * g / sqrt(var + eps) per output channel, b' = beta - mean * g / sqrt(var + eps) (in fp64, rounded
//               once); bn0 cannot be folded (the zero padding of conv1 comes after it) and is one elementwise launch.
//   Spectral    torchlibrosa's Spectrogram + LogmelFilterBank as Cnn14 holds them (audio.py:123-131: n_fft = window 1024,
//               hop 320, centre + reflect padding, power 2, 64 mel bins, 10 log10 with amin 1e-10) and TRANSFORMS_16000
//               (ldm/data/extract_mel_spectrogram.py:15-38,140-150; audio-chatgpt.py:468-491: librosa.stft 1024 / hop 256,
//               magnitude, 80 slaney mel bins, log10 / scale / clip).  Both are: pad -> framed DFT as ONE GEMM whose A rows
//               are overlapping windows of the padded signal (row pitch = hop, K = n_fft; the analysis window is folded
//               into the basis) -> re^2 + im^2 (-> sqrt) -> mel GEMM -> log epilogue.
//   Resampler   torchaudio.transforms.Resample as CLAPWrapper.resample_and_duration uses it (CLAPWrapper.py:103-110): its
//               polyphase sinc kernel bank [new, 2 width + orig] applied with stride orig -- the same overlapping-row GEMM.
// The front ends always run in exact fp32 (they are a few GFLOP per clip and feed a logarithm).
#include "models.h"

#include <cmath>
#include <cstring>

namespace maa {

void launch_pad1d(const Ctx& ctx, const float* x, int B, int n, int left, int total, int ldo, int mode, float* out);
void launch_spec_power(const Ctx& ctx, const float* y, long long rows, int nf, int ldy, int ldm, int power2, float* mag);
void launch_logmel(const Ctx& ctx, const float* mel, int B, int frames, int n_mels, int ldmel, int kind, float amin,
                   float ref_db, int layout, float* out);
void launch_affine_lastdim(const Ctx& ctx, const float* x, long long n, int F, const float* scale, const float* shift,
                           float* out);
void launch_cnn14_pool(const Ctx& ctx, const float* x, int B, int T, int F, int C, float* out);
void launch_l2norm_rows(const Ctx& ctx, const float* x, int B, int C, float* out);
void launch_gelu(const Ctx& ctx, const float* x, long long n, float* out);

// ------------------------------------------------------------------------------------------ Cnn14 + Projection
struct ClapAudio::Impl {
    maa_clap_audio_config cfg;
    int precision = 0;
    WeightStore ws;
    explicit Impl(int prec) : precision(prec), ws(prec != 0) {}

    struct Block {
        PackedW c1, c2;
    };
    std::vector<Block> blocks;
    float *bn0_scale = nullptr, *bn0_shift = nullptr;
    PackedW fc1, proj1, proj2;
    float *proj_g = nullptr, *proj_b = nullptr;

    // conv (no bias) followed by eval-mode BatchNorm -> one conv with bias
    PackedW pack_conv_bn(const StateDict& sd, const std::string& conv, const std::string& bn) {
        const HostTensor& w = get(sd, conv + ".weight");
        const HostTensor &g = get(sd, bn + ".weight"), &b = get(sd, bn + ".bias"), &mu = get(sd, bn + ".running_mean"),
                         &var = get(sd, bn + ".running_var");
        MAA_CHECK(w.shape.size() == 4 && w.shape[2] == 3 && w.shape[3] == 3, conv + ": 3x3 convolution expected");
        const int Cout = (int)w.shape[0];
        const long long per = w.numel() / Cout;
        MAA_CHECK(g.numel() == Cout && b.numel() == Cout && mu.numel() == Cout && var.numel() == Cout, bn + ": BatchNorm width");
        std::vector<float> wf((size_t)w.numel()), bf((size_t)Cout);
        for (int co = 0; co < Cout; ++co) {
            const double s = (double)g.data[co] / std::sqrt((double)var.data[co] + (double)cfg.bn_eps);
            for (long long i = 0; i < per; ++i) wf[(size_t)(co * per + i)] = (float)((double)w.data[co * per + i] * s);
            bf[(size_t)co] = (float)((double)b.data[co] - (double)mu.data[co] * s);
        }
        StateDict tmp;
        HostTensor hw, hb;
        hw.data = wf.data();
        hw.shape = w.shape;
        hb.data = bf.data();
        hb.shape = {Cout};
        tmp["w"] = hw;
        tmp["b"] = hb;
        return ws.pack_conv(tmp, "w", "b", 3, 3);
    }

    void build(const StateDict& sd) {
        const int F = cfg.mel_bins;
        {
            const HostTensor &g = get(sd, "base.bn0.weight"), &b = get(sd, "base.bn0.bias"), &mu = get(sd, "base.bn0.running_mean"),
                             &var = get(sd, "base.bn0.running_var");
            MAA_CHECK(g.numel() == F && b.numel() == F && mu.numel() == F && var.numel() == F, "bn0 runs over the mel bins");
            std::vector<float> sc((size_t)F), sh((size_t)F);
            for (int f = 0; f < F; ++f) {
                // kept as the two-step form of ATen's eval-mode batch_norm: (x - mean) * invstd * g + b == x * s + (b - mean * s)
                const double s = (double)g.data[f] / std::sqrt((double)var.data[f] + (double)cfg.bn_eps);
                sc[(size_t)f] = (float)s;
                sh[(size_t)f] = (float)((double)b.data[f] - (double)mu.data[f] * s);
            }
            bn0_scale = ws.upload(sc);
            bn0_shift = ws.upload(sh);
        }
        int cin = 1;
        for (int i = 0; i < cfg.n_blocks; ++i) {
            const std::string p = "base.conv_block" + std::to_string(i + 1) + ".";
            Block blk;
            blk.c1 = pack_conv_bn(sd, p + "conv1", p + "bn1");
            blk.c2 = pack_conv_bn(sd, p + "conv2", p + "bn2");
            MAA_CHECK(blk.c1.K == 9 * cin && blk.c1.N == cfg.channels[i] && blk.c2.K == 9 * cfg.channels[i] &&
                          blk.c2.N == cfg.channels[i],
                      p + ": channel widths differ from the configuration");
            cin = cfg.channels[i];
            blocks.push_back(blk);
        }
        fc1 = ws.pack_conv(sd, "base.fc1.weight", "base.fc1.bias", 1, 1);
        MAA_CHECK(fc1.K == cin && fc1.N == cfg.out_emb, "fc1 shape");
        proj1 = ws.pack_conv(sd, "projection.linear1.weight", "", 1, 1);
        proj2 = ws.pack_conv(sd, "projection.linear2.weight", "", 1, 1);
        proj_g = ws.vec(sd, "projection.layer_norm.weight");
        proj_b = ws.vec(sd, "projection.layer_norm.bias");
        MAA_CHECK(proj1.K == cfg.out_emb && proj1.N == cfg.d_proj && proj2.K == cfg.d_proj && proj2.N == cfg.d_proj,
                  "CLAP audio projection widths");
    }

    void embed(Ctx& ctx, const float* logmel, int B, int T, float* d_emb, float* d_z) {
        int H = T, W = cfg.mel_bins;
        // [B, 1, T, F] is [B, T, F, 1] channels-last
        T4 x = alloc_t(ctx, B, H, W, 1);
        launch_affine_lastdim(ctx, logmel, (long long)B * H * W, W, bn0_scale, bn0_shift, x.p);
        for (int i = 0; i < cfg.n_blocks; ++i) {
            const int C = cfg.channels[i];
            ConvOpt co;
            co.KH = co.KW = 3;
            co.pad = 1;
            co.act = 2;                                   // ReLU after the folded BatchNorm (audio.py:36-37)
            T4 a = alloc_t(ctx, B, H, W, C), b = alloc_t(ctx, B, H, W, C);
            conv_into(ctx, x, nullptr, blocks[(size_t)i].c1, co, a);
            conv_into(ctx, a, nullptr, blocks[(size_t)i].c2, co, b);
            if (i + 1 < cfg.n_blocks) {                   // avg_pool2d(2) floors odd sizes; the last block pools (1, 1)
                MAA_CHECK(H >= 2 && W >= 2, "clip too short for Cnn14's five poolings");
                T4 pz = alloc_t(ctx, B, H / 2, W / 2, C);
                launch_avgpool2(ctx, b.p, B, H, W, C, pz.p);
                x = pz;
                H /= 2;
                W /= 2;
            } else {
                x = b;
            }
        }
        const int C = cfg.channels[cfg.n_blocks - 1], E = cfg.out_emb, D = cfg.d_proj;
        float* pooled = ctx.ws.alloc_f((size_t)B * C);
        launch_cnn14_pool(ctx, x.p, B, H, W, C, pooled);
        float* emb = d_emb ? d_emb : ctx.ws.alloc_f((size_t)B * E);
        linear_into(ctx, pooled, C, B, C, fc1, nullptr, 0, emb, E, 0, 0, 0, 0, /*act=*/2);
        float* e1 = ctx.ws.alloc_f((size_t)B * D);
        float* g = ctx.ws.alloc_f((size_t)B * D);
        float* e2 = ctx.ws.alloc_f((size_t)B * D);
        float* ln = ctx.ws.alloc_f((size_t)B * D);
        linear_into(ctx, emb, E, B, E, proj1, nullptr, 0, e1, D);
        launch_gelu(ctx, e1, (long long)B * D, g);
        linear_into(ctx, g, D, B, D, proj2, e1, D, e2, D);
        launch_layernorm(ctx, e2, B, D, proj_g, proj_b, 1e-5f, ln);
        launch_l2norm_rows(ctx, ln, B, D, d_z);
    }
};

ClapAudio::ClapAudio(const maa_clap_audio_config& cfg, const StateDict& sd, int precision) : impl_(new Impl(precision)) {
    impl_->cfg = cfg;
    try {
        impl_->build(sd);
    } catch (...) {
        delete impl_;
        throw;
    }
}
ClapAudio::~ClapAudio() { delete impl_; }
const maa_clap_audio_config& ClapAudio::config() const { return impl_->cfg; }
void ClapAudio::embed(Ctx& ctx, const float* d_logmel, int B, int T, float* d_embedding, float* d_z) {
    PrecisionGuard guard(ctx, impl_->precision);
    run_sized(ctx, [&] { impl_->embed(ctx, d_logmel, B, T, d_embedding, d_z); });
}

// ------------------------------------------------------------------------------------------ log-mel front end
struct Spectral::Impl {
    maa_spectral_config cfg;
    WeightStore ws{false};        // exact-fp32 layout
    PackedW basis, melw;
    int ldm = 0;                  // pitch of the magnitude rows = K of the mel product (n_freq rounded up to 4)
};

Spectral::Spectral(const maa_spectral_config& cfg, const float* h_basis, const float* h_melw) : impl_(new Impl) {
    impl_->cfg = cfg;
    try {
        const int nf = cfg.n_freq;
        impl_->ldm = (nf + 3) / 4 * 4;
        StateDict tmp;
        HostTensor hb;
        hb.data = h_basis;
        hb.shape = {2LL * nf, cfg.n_fft};
        tmp["basis"] = hb;
        impl_->basis = impl_->ws.pack_conv(tmp, "basis", "", 1, 1);
        std::vector<float> mw((size_t)cfg.n_mels * impl_->ldm, 0.f);
        for (int m = 0; m < cfg.n_mels; ++m)
            std::memcpy(mw.data() + (size_t)m * impl_->ldm, h_melw + (size_t)m * nf, sizeof(float) * nf);
        HostTensor hm;
        hm.data = mw.data();
        hm.shape = {cfg.n_mels, impl_->ldm};
        tmp["melw"] = hm;
        impl_->melw = impl_->ws.pack_conv(tmp, "melw", "", 1, 1);
    } catch (...) {
        delete impl_;
        throw;
    }
}
Spectral::~Spectral() { delete impl_; }
const maa_spectral_config& Spectral::config() const { return impl_->cfg; }

void Spectral::forward(Ctx& ctx, const float* d_wav, int B, int n, float* d_out) {
    const maa_spectral_config& c = impl_->cfg;
    const int nf = c.n_freq, half = c.n_fft / 2;
    const int frames = 1 + n / c.hop;
    // padded rows are 16-byte aligned so that every sample's overlapping-window matrix takes the aligned gather
    const int total = n + 2 * half, ldp = (total + 3) / 4 * 4;
    PrecisionGuard guard(ctx, 0);
    run_sized(ctx, [&] {
        float* xp = ctx.ws.alloc_f((size_t)B * ldp);
        launch_pad1d(ctx, d_wav, B, n, half, total, ldp, c.pad_mode, xp);
        const int ldy = 2 * nf;
        float* y = ctx.ws.alloc_f((size_t)B * frames * ldy);
        for (int b = 0; b < B; ++b)      // A[t, k] = xp[b, t hop + k]: row pitch hop < K, rows overlap (read-only)
            linear_into(ctx, xp + (size_t)b * ldp, c.hop, frames, c.n_fft, impl_->basis, nullptr, 0,
                        y + (size_t)b * frames * ldy, ldy);
        float* mag = ctx.ws.alloc_f((size_t)B * frames * impl_->ldm);
        launch_spec_power(ctx, y, (long long)B * frames, nf, ldy, impl_->ldm, c.power == 2 ? 1 : 0, mag);
        float* mel = ctx.ws.alloc_f((size_t)B * frames * c.n_mels);
        linear_into(ctx, mag, impl_->ldm, (long long)B * frames, impl_->ldm, impl_->melw, nullptr, 0, mel, c.n_mels);
        const float ref_db = 10.0f * std::log10(std::fmax(c.amin, c.ref));
        launch_logmel(ctx, mel, B, frames, c.n_mels, c.n_mels, c.log_kind, c.amin, ref_db, c.out_layout, d_out);
    });
}

// ------------------------------------------------------------------------------------------ resampler
struct Resampler::Impl {
    int orig = 1, neu = 1, width = 0, klen = 0, kpad = 0;
    WeightStore ws{false};
    PackedW bank;
};

Resampler::Resampler(int orig, int neu, int width, int klen, const float* h_kernels) : impl_(new Impl) {
    impl_->orig = orig;
    impl_->neu = neu;
    impl_->width = width;
    impl_->klen = klen;
    impl_->kpad = (klen + 3) / 4 * 4;
    try {
        MAA_CHECK(klen == 2 * width + orig, "resampler kernel length must be 2 width + orig_freq");
        std::vector<float> kw((size_t)neu * impl_->kpad, 0.f);
        for (int p = 0; p < neu; ++p) std::memcpy(kw.data() + (size_t)p * impl_->kpad, h_kernels + (size_t)p * klen, sizeof(float) * klen);
        StateDict tmp;
        HostTensor h;
        h.data = kw.data();
        h.shape = {neu, impl_->kpad};
        tmp["k"] = h;
        impl_->bank = impl_->ws.pack_conv(tmp, "k", "", 1, 1);
    } catch (...) {
        delete impl_;
        throw;
    }
}
Resampler::~Resampler() { delete impl_; }
long long Resampler::out_length(long long n) const {
    return ((long long)impl_->neu * n + impl_->orig - 1) / impl_->orig;           // ceil(new * length / orig)
}

void Resampler::forward(Ctx& ctx, const float* d_wav, int B, int n, float* d_out) {
    const Impl& r = *impl_;
    // (a row pitch orig that is not a multiple of 4 takes the engine's element-wise gather instead of the aligned one)
    const int frames = n / r.orig + 1;                      // conv1d over pad(width, width + orig) with stride orig
    const int total = n + 2 * r.width + r.orig;
    const int ldp = (total + (r.kpad - r.klen) + 3) / 4 * 4;   // the zero-weight taps past klen still read memory
    const long long target = out_length(n);
    MAA_CHECK((long long)frames * r.neu >= target, "resampler frame count");
    PrecisionGuard guard(ctx, 0);
    run_sized(ctx, [&] {
        float* xp = ctx.ws.alloc_f((size_t)B * ldp);
        launch_pad1d(ctx, d_wav, B, n, r.width, total, ldp, 0, xp);
        float* y = ctx.ws.alloc_f((size_t)B * frames * r.neu);
        for (int b = 0; b < B; ++b)
            linear_into(ctx, xp + (size_t)b * ldp, r.orig, frames, r.kpad, r.bank, nullptr, 0, y + (size_t)b * frames * r.neu, r.neu);
        // [B, frames, new] is the resampled signal in time order; keep the first ceil(new n / orig) samples of each row
        if (!ctx.ws.dry)
            MAA_HIP(hipMemcpy2DAsync(d_out, (size_t)target * sizeof(float), y, (size_t)frames * r.neu * sizeof(float),
                                     (size_t)target * sizeof(float), (size_t)B, hipMemcpyDeviceToDevice, ctx.stream));
    });
}

}  // namespace maa
