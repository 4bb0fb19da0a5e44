// Mel VAE (AutoencoderKL) decoder / encoder executor on channels-last fp32 activations.
//
// Mirrors (relative to text_to_audio/Make_An_Audio in the reference):
//   ldm/modules/diffusionmodules/model.py:462-568 Decoder, :368-459 Encoder, :82-141 ResnetBlock (temb=None),
//   :150-202 AttnBlock (single head, scale C^-1/2), :42-57 Upsample, :60-79 Downsample (pad right/bottom)
//   ldm/models/autoencoder.py:345-354 quant_conv / post_quant_conv
// q, k, v 1x1 convs of an AttnBlock run as one GEMM (N = 3C); nearest-2x upsampling is folded into the
// following conv's gather; GroupNorm(eps 1e-6)+swish is one kernel; residual adds are igemm epilogues.
#include "models.h"

#include <cmath>

namespace maa {

namespace {
struct VResW {
    int cin = 0, cout = 0;
    float *g1, *b1, *g2, *b2;
    PackedW conv1, conv2, nin;
    bool has_nin = false;
};
struct VAttnW {
    int c = 0;
    float *ng, *nb;
    PackedW qkv, proj;
};
struct VLevel {
    std::vector<VResW> blocks;
    std::vector<VAttnW> attns;
    bool has_resample = false;
    PackedW resample, resample_up2;
    int resample_c = 0;
};
}  // namespace

struct VAE::Impl {
    maa_vae_config cfg;
    int precision = 0;
    WeightStore ws;
    explicit Impl(int prec) : precision(prec), ws(prec != 0) {}
    bool has_encoder = false;
    // decoder
    PackedW d_conv_in, d_conv_out, post_quant;
    VResW d_mid1, d_mid2;
    VAttnW d_mid_attn;
    std::vector<VLevel> d_up;   // indexed by level
    float *d_ng = nullptr, *d_nb = nullptr;
    int d_block_in_top = 0, d_last_c = 0;
    // encoder
    PackedW e_conv_in, e_conv_out, quant;
    VResW e_mid1, e_mid2;
    VAttnW e_mid_attn;
    std::vector<VLevel> e_down;
    float *e_ng = nullptr, *e_nb = nullptr;

    VResW load_res(const StateDict& sd, const std::string& p, int cin, int cout) {
        VResW r;
        r.cin = cin;
        r.cout = cout;
        r.g1 = ws.vec(sd, p + "norm1.weight");
        r.b1 = ws.vec(sd, p + "norm1.bias");
        r.conv1 = ws.pack_conv(sd, p + "conv1.weight", p + "conv1.bias", 3, 3);
        r.g2 = ws.vec(sd, p + "norm2.weight");
        r.b2 = ws.vec(sd, p + "norm2.bias");
        r.conv2 = ws.pack_conv(sd, p + "conv2.weight", p + "conv2.bias", 3, 3);
        r.has_nin = has(sd, p + "nin_shortcut.weight");
        MAA_CHECK(r.has_nin == (cin != cout), "nin_shortcut presence " + p);
        if (r.has_nin) r.nin = ws.pack_conv(sd, p + "nin_shortcut.weight", p + "nin_shortcut.bias", 1, 1);
        return r;
    }
    VAttnW load_attn(const StateDict& sd, const std::string& p, int c) {
        VAttnW a;
        a.c = c;
        a.ng = ws.vec(sd, p + "norm.weight");
        a.nb = ws.vec(sd, p + "norm.bias");
        a.qkv = ws.pack_concat(sd, {p + "q.weight", p + "k.weight", p + "v.weight"},
                               {p + "q.bias", p + "k.bias", p + "v.bias"});
        a.proj = ws.pack_conv(sd, p + "proj_out.weight", p + "proj_out.bias", 1, 1);
        return a;
    }
    bool attn_at(int res) const {
        for (int i = 0; i < cfg.n_attn_resolutions; ++i)
            if (cfg.attn_resolutions[i] == res) return true;
        return false;
    }

    void build(const StateDict& sd) {
        const int nres = cfg.n_ch_mult, ch = cfg.ch;
        // ---- decoder (model.py:462-533)
        std::string p = "decoder.";
        int block_in = ch * cfg.ch_mult[nres - 1];
        int curr = cfg.resolution >> (nres - 1);
        d_block_in_top = block_in;
        d_conv_in = ws.pack_conv(sd, p + "conv_in.weight", p + "conv_in.bias", 3, 3);
        d_mid1 = load_res(sd, p + "mid.block_1.", block_in, block_in);
        d_mid_attn = load_attn(sd, p + "mid.attn_1.", block_in);
        d_mid2 = load_res(sd, p + "mid.block_2.", block_in, block_in);
        d_up.resize(nres);
        for (int lvl = nres - 1; lvl >= 0; --lvl) {
            const int block_out = ch * cfg.ch_mult[lvl];
            VLevel& L = d_up[lvl];
            for (int ib = 0; ib < cfg.num_res_blocks + 1; ++ib) {
                const std::string q = p + "up." + std::to_string(lvl) + ".";
                L.blocks.push_back(load_res(sd, q + "block." + std::to_string(ib) + ".", block_in, block_out));
                block_in = block_out;
                if (attn_at(curr)) L.attns.push_back(load_attn(sd, q + "attn." + std::to_string(ib) + ".", block_in));
            }
            if (lvl != 0) {
                L.has_resample = true;
                L.resample_c = block_in;
                L.resample = ws.pack_conv(sd, p + "up." + std::to_string(lvl) + ".upsample.conv.weight",
                                          p + "up." + std::to_string(lvl) + ".upsample.conv.bias", 3, 3);
                L.resample_up2 = ws.pack_conv_up2(sd, p + "up." + std::to_string(lvl) + ".upsample.conv.weight",
                                                  p + "up." + std::to_string(lvl) + ".upsample.conv.bias");
                curr *= 2;
            }
        }
        d_last_c = block_in;
        d_ng = ws.vec(sd, p + "norm_out.weight");
        d_nb = ws.vec(sd, p + "norm_out.bias");
        d_conv_out = ws.pack_conv(sd, p + "conv_out.weight", p + "conv_out.bias", 3, 3);
        post_quant = ws.pack_conv(sd, "post_quant_conv.weight", "post_quant_conv.bias", 1, 1);
        // ---- encoder (model.py:368-432), optional
        has_encoder = has(sd, "encoder.conv_in.weight");
        if (!has_encoder) return;
        p = "encoder.";
        curr = cfg.resolution;
        e_conv_in = ws.pack_conv(sd, p + "conv_in.weight", p + "conv_in.bias", 3, 3);
        e_down.resize(nres);
        block_in = ch;
        for (int lvl = 0; lvl < nres; ++lvl) {
            block_in = ch * (lvl == 0 ? 1 : cfg.ch_mult[lvl - 1]);
            const int block_out = ch * cfg.ch_mult[lvl];
            VLevel& L = e_down[lvl];
            const std::string q = p + "down." + std::to_string(lvl) + ".";
            for (int ib = 0; ib < cfg.num_res_blocks; ++ib) {
                L.blocks.push_back(load_res(sd, q + "block." + std::to_string(ib) + ".", block_in, block_out));
                block_in = block_out;
                if (attn_at(curr)) L.attns.push_back(load_attn(sd, q + "attn." + std::to_string(ib) + ".", block_in));
            }
            if (lvl != nres - 1) {
                L.has_resample = true;
                L.resample_c = block_in;
                L.resample = ws.pack_conv(sd, q + "downsample.conv.weight", q + "downsample.conv.bias", 3, 3);
                curr /= 2;
            }
        }
        e_mid1 = load_res(sd, p + "mid.block_1.", block_in, block_in);
        e_mid_attn = load_attn(sd, p + "mid.attn_1.", block_in);
        e_mid2 = load_res(sd, p + "mid.block_2.", block_in, block_in);
        e_ng = ws.vec(sd, p + "norm_out.weight");
        e_nb = ws.vec(sd, p + "norm_out.bias");
        e_conv_out = ws.pack_conv(sd, p + "conv_out.weight", p + "conv_out.bias", 3, 3);
        quant = ws.pack_conv(sd, "quant_conv.weight", "quant_conv.bias", 1, 1);
    }

    T4 run_res(Ctx& ctx, const VResW& r, const T4& x) {
        T4 out = alloc_t(ctx, x.B, x.H, x.W, r.cout);
        const size_t mk = ctx.ws.mark();
        T4 t1 = alloc_t(ctx, x.B, x.H, x.W, r.cin);
        t1.split = split_for_gemm(ctx, r.cin);
        launch_groupnorm(ctx, x.p, r.cin, r.cin, nullptr, 0, 0, x.B, x.H * x.W, 32, r.g1, r.b1, 1e-6f, 1, t1.p, t1.split);
        T4 h1 = alloc_t(ctx, x.B, x.H, x.W, r.cout);
        ConvOpt o;
        o.KH = o.KW = 3;
        o.pad = 1;
        conv_into(ctx, t1, nullptr, r.conv1, o, h1);
        T4 t2 = alloc_t(ctx, x.B, x.H, x.W, r.cout);
        t2.split = split_for_gemm(ctx, r.cout);
        launch_groupnorm(ctx, h1.p, r.cout, r.cout, nullptr, 0, 0, x.B, x.H * x.W, 32, r.g2, r.b2, 1e-6f, 1, t2.p, t2.split);
        const float* resid = x.p;
        if (r.has_nin) {
            T4 sk = alloc_t(ctx, x.B, x.H, x.W, r.cout);
            ConvOpt os;
            conv_into(ctx, x, nullptr, r.nin, os, sk);
            resid = sk.p;
        }
        ConvOpt o2 = o;
        o2.res = resid;
        conv_into(ctx, t2, nullptr, r.conv2, o2, out);
        ctx.ws.release(mk);
        return out;
    }

    T4 run_attn(Ctx& ctx, const VAttnW& a, const T4& x) {
        const int C = a.c, HW = x.H * x.W;
        const long long M = (long long)x.B * HW;
        T4 out = alloc_t(ctx, x.B, x.H, x.W, C);
        const size_t mk = ctx.ws.mark();
        float* xn = ctx.ws.alloc_f((size_t)M * C);
        const bool sp = split_for_gemm(ctx, C);
        launch_groupnorm(ctx, x.p, C, C, nullptr, 0, 0, x.B, HW, 32, a.ng, a.nb, 1e-6f, 0, xn, sp);
        float* qkv = ctx.ws.alloc_f((size_t)M * 3 * C);
        linear_into(ctx, xn, C, M, C, a.qkv, nullptr, 0, qkv, 3 * C, 0, 0, sp ? M : 0);
        float* o = ctx.ws.alloc_f((size_t)M * C);
        // w = softmax(q k^T * C^-1/2) over keys; h = w v   (model.py:186-198)
        const float sc = (float)std::pow((double)(int)C, -0.5);
        attention_into(ctx, qkv, 3 * C, 0, qkv + C, 3 * C, 0, qkv + 2 * C, 3 * C, 0, x.B, 1, C, HW, HW, sc, o, C);
        linear_into(ctx, o, C, M, C, a.proj, x.p, C, out.p, C);
        ctx.ws.release(mk);
        return out;
    }

    // spec = true: the tools' next line folded in -- clamp((mel + 1) / 2, 0, 1) (audio-chatgpt.py:175-176), written as [B, H, W]
    void decode(Ctx& ctx, const float* z_nchw, int B, int h, int w, float inv_scale, float* mel_nchw, bool spec = false) {
        const int nres = cfg.n_ch_mult;
        T4 z = alloc_t(ctx, B, h, w, cfg.embed_dim);
        launch_nchw_to_nhwc(ctx, z_nchw, B, cfg.embed_dim, h * w, z.p);
        if (inv_scale != 1.0f) launch_scale(ctx, z.p, z.numel(), inv_scale, z.p);   // z / scale_factor
        T4 zq = alloc_t(ctx, B, h, w, cfg.z_channels);
        ConvOpt o1;
        conv_into(ctx, z, nullptr, post_quant, o1, zq);
        T4 hcur = alloc_t(ctx, B, h, w, d_block_in_top);
        ConvOpt o3;
        o3.KH = o3.KW = 3;
        o3.pad = 1;
        conv_into(ctx, zq, nullptr, d_conv_in, o3, hcur);
        hcur = run_res(ctx, d_mid1, hcur);
        hcur = run_attn(ctx, d_mid_attn, hcur);
        hcur = run_res(ctx, d_mid2, hcur);
        for (int lvl = nres - 1; lvl >= 0; --lvl) {
            VLevel& L = d_up[lvl];
            for (size_t ib = 0; ib < L.blocks.size(); ++ib) {
                hcur = run_res(ctx, L.blocks[ib], hcur);
                if (!L.attns.empty()) hcur = run_attn(ctx, L.attns[ib], hcur);
            }
            if (L.has_resample) {
                T4 up = alloc_t(ctx, B, hcur.H * 2, hcur.W * 2, L.resample_c);
                if (!conv_up2_into(ctx, hcur, L.resample_up2, up)) {
                    ConvOpt ou = o3;
                    ou.up = 1;
                    conv_into(ctx, hcur, nullptr, L.resample, ou, up);
                }
                hcur = up;
            }
        }
        T4 hn = alloc_t(ctx, B, hcur.H, hcur.W, d_last_c);
        hn.split = split_for_gemm(ctx, d_last_c);
        launch_groupnorm(ctx, hcur.p, d_last_c, d_last_c, nullptr, 0, 0, B, hcur.H * hcur.W, 32, d_ng, d_nb, 1e-6f, 1,
                         hn.p, hn.split);
        T4 out = alloc_t(ctx, B, hcur.H, hcur.W, cfg.out_ch);
        conv_into(ctx, hn, nullptr, d_conv_out, o3, out);
        if (spec) {
            MAA_CHECK(cfg.out_ch == 1, "decode_spec: the mel decoder has one output channel");
            launch_spec_from_mel(ctx, out.p, (long long)B * hcur.H * hcur.W, mel_nchw);
        } else {
            launch_nhwc_to_nchw(ctx, out.p, B, cfg.out_ch, hcur.H * hcur.W, mel_nchw, cfg.out_ch);
        }
    }

    void encode(Ctx& ctx, const float* mel_nchw, int B, int H, int W, float* moments_nchw) {
        MAA_CHECK(has_encoder, "VAE was created without encoder weights");
        const int nres = cfg.n_ch_mult;
        T4 x = alloc_t(ctx, B, H, W, cfg.in_channels);
        launch_nchw_to_nhwc(ctx, mel_nchw, B, cfg.in_channels, H * W, x.p);
        ConvOpt o3;
        o3.KH = o3.KW = 3;
        o3.pad = 1;
        T4 hcur = alloc_t(ctx, B, H, W, cfg.ch);
        conv_into(ctx, x, nullptr, e_conv_in, o3, hcur);
        for (int lvl = 0; lvl < nres; ++lvl) {
            VLevel& L = e_down[lvl];
            for (size_t ib = 0; ib < L.blocks.size(); ++ib) {
                hcur = run_res(ctx, L.blocks[ib], hcur);
                if (!L.attns.empty()) hcur = run_attn(ctx, L.attns[ib], hcur);
            }
            if (L.has_resample) {   // pad (0,1,0,1) then conv3x3 stride 2 pad 0 (model.py:72-77)
                const int Ho = (hcur.H + 1 - 3) / 2 + 1, Wo = (hcur.W + 1 - 3) / 2 + 1;
                T4 dn = alloc_t(ctx, B, Ho, Wo, L.resample_c);
                ConvOpt od;
                od.KH = od.KW = 3;
                od.stride = 2;
                od.pad = 0;
                od.pad_h = 0;
                conv_into(ctx, hcur, nullptr, L.resample, od, dn);
                hcur = dn;
            }
        }
        hcur = run_res(ctx, e_mid1, hcur);
        hcur = run_attn(ctx, e_mid_attn, hcur);
        hcur = run_res(ctx, e_mid2, hcur);
        T4 hn = alloc_t(ctx, B, hcur.H, hcur.W, hcur.C);
        hn.split = split_for_gemm(ctx, hcur.C);
        launch_groupnorm(ctx, hcur.p, hcur.C, hcur.C, nullptr, 0, 0, B, hcur.H * hcur.W, 32, e_ng, e_nb, 1e-6f, 1, hn.p,
                         hn.split);
        const int oc = (cfg.double_z ? 2 : 1) * cfg.z_channels;
        T4 eo = alloc_t(ctx, B, hcur.H, hcur.W, oc);
        conv_into(ctx, hn, nullptr, e_conv_out, o3, eo);
        T4 mo = alloc_t(ctx, B, hcur.H, hcur.W, 2 * cfg.embed_dim);
        ConvOpt o1;
        conv_into(ctx, eo, nullptr, quant, o1, mo);
        launch_nhwc_to_nchw(ctx, mo.p, B, 2 * cfg.embed_dim, hcur.H * hcur.W, moments_nchw, 2 * cfg.embed_dim);
    }
};

VAE::VAE(const maa_vae_config& cfg, const StateDict& sd, int precision) : impl_(new Impl(precision)) {
    impl_->cfg = cfg;
    impl_->build(sd);
}
VAE::~VAE() { delete impl_; }
const maa_vae_config& VAE::config() const { return impl_->cfg; }

void VAE::decode(Ctx& ctx, const float* z, int B, int h, int w, float inv_scale, float* mel) {
    PrecisionGuard pg(ctx, impl_->precision);
    run_sized(ctx, [&] { impl_->decode(ctx, z, B, h, w, inv_scale, mel); });
}
void VAE::decode_spec(Ctx& ctx, const float* z, int B, int h, int w, float inv_scale, float* spec) {
    PrecisionGuard pg(ctx, impl_->precision);
    run_sized(ctx, [&] { impl_->decode(ctx, z, B, h, w, inv_scale, spec, true); });
}
void VAE::encode_moments(Ctx& ctx, const float* mel, int B, int H, int W, float* moments) {
    PrecisionGuard pg(ctx, impl_->precision);
    run_sized(ctx, [&] { impl_->encode(ctx, mel, B, H, W, moments); });
}

}  // namespace maa
