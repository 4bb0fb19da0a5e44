// HiFi-GAN / BigVGAN MRF generator executor on channels-last sequences [B, L, C].
//
// Mirrors: NeuralSeq/modules/hifigan/hifigan.py:104-178 (HifiGanGenerator, f0=None), :30-67 ResBlock1, :70-91 ResBlock2
//          text_to_audio/Make_An_Audio/vocoder/hifigan/modules.py:86-136 (same graph)
//          text_to_audio/Make_An_Audio/vocoder/bigvgan/models.py:30-81,133-203 (AMPBlock1, BigVGAN), :90-132 AMPBlock2
// What is fused instead of launched (same maths):
//   * every leaky-ReLU is applied while the following conv stages its A tile
//   * `xt + x` and the MRF mean (rb0+rb1+rb2)/3 are igemm epilogues (out_scale 1/3, accumulate)
//   * ConvTranspose1d runs as two polyphase GEMMs writing interleaved output rows (runtime.cpp)
//   * weight-norm (weight_g, weight_v) is folded at load exactly as remove_weight_norm does
//   * NSF branch (hifigan.py:111-132, 145-157): the harmonic source comes from nsf.hip; each stage's strided
//     noise_convs[i](har_source) is one more implicit GEMM that accumulates into the ups[i] output (x = x + x_source)
#include "models.h"

#include <cmath>
#include <cstring>

namespace maa {

namespace {
struct ConvK {
    PackedW w;
    int k = 1, dil = 1;
};
struct ResBlockW {
    std::vector<ConvK> c1, c2;      // resblock "2" (ResBlock2 / AMPBlock2): c2 is empty, c1 holds `convs.{m}`
    // BigVGAN: per-activation snake parameters (device), 2 per (c1, c2) pair (one per conv of an AMPBlock2)
    std::vector<float*> alpha, inv_beta;
};
struct UpW {
    PackedW ph[2];
    int r_start[2] = {0, 0}, r_count[2] = {0, 0};
    int n_groups = 0;
    int stride = 1, k = 1, cin = 0, cout = 0;
};

// fold weight-norm pairs: w = g * v / ||v||_(dims != 0)
struct Folded {
    std::vector<std::vector<float>> storage;
    StateDict sd;
};
void fold_weight_norm(const StateDict& in, Folded& out) {
    for (auto& kv : in) {
        const std::string& k = kv.first;
        auto ends = [&](const char* suf) {
            const size_t n = std::strlen(suf);
            return k.size() >= n && k.compare(k.size() - n, n, suf) == 0;
        };
        if (ends(".weight_g")) {
            const std::string base = k.substr(0, k.size() - 9);
            const HostTensor& g = kv.second;
            const HostTensor& v = get(in, base + ".weight_v");
            const long long rows = v.shape[0], per = v.numel() / rows;
            MAA_CHECK(g.numel() == rows, "weight_g shape " + k);
            std::vector<float> w((size_t)v.numel());
            for (long long r = 0; r < rows; ++r) {
                // torch.norm_except_dim accumulates in fp32; do the same
                float s = 0.f;
                for (long long i = 0; i < per; ++i) s += v.data[r * per + i] * v.data[r * per + i];
                const float scale = g.data[r] / std::sqrt(s);
                for (long long i = 0; i < per; ++i) w[(size_t)(r * per + i)] = v.data[r * per + i] * scale;
            }
            out.storage.push_back(std::move(w));
            HostTensor t;
            t.data = out.storage.back().data();
            t.shape = v.shape;
            out.sd[base + ".weight"] = t;
        } else if (!ends(".weight_v")) {
            out.sd[k] = kv.second;
        }
    }
}
}  // namespace

struct Vocoder::Impl {
    maa_vocoder_config cfg;
    int precision = 0;
    WeightStore ws;
    explicit Impl(int prec) : precision(prec), ws(prec != 0) {}
    PackedW conv_pre, conv_post;
    std::vector<UpW> ups;
    std::vector<ResBlockW> rbs;
    float *post_alpha = nullptr, *post_inv_beta = nullptr;
    int hop = 1;
    // NSF branch
    std::vector<PackedW> noise_convs;
    std::vector<int> noise_stride;
    float *src_w = nullptr, *src_b = nullptr;     // m_source.l_linear [1, harmonics+1], [1]

    void snake_params(const StateDict& sd, const std::string& p, float** alpha, float** inv_beta) {
        // activations.py:107-119: alpha, beta = exp(param) if logscale; x + 1/(beta + 1e-9) * sin^2(alpha x)
        const HostTensor& a = get(sd, p + "act.alpha");
        const HostTensor& b = cfg.snake_beta ? get(sd, p + "act.beta") : a;
        std::vector<float> ha((size_t)a.numel()), hb((size_t)a.numel());
        for (long long i = 0; i < a.numel(); ++i) {
            float av = a.data[i], bv = b.data[i];
            if (cfg.snake_logscale) {
                av = std::exp(av);
                bv = std::exp(bv);
            }
            ha[(size_t)i] = av;
            hb[(size_t)i] = 1.0f / (bv + 1e-9f);
        }
        *alpha = ws.upload(ha);
        *inv_beta = ws.upload(hb);
    }

    void build(const StateDict& raw) {
        Folded f;
        fold_weight_norm(raw, f);
        const StateDict& sd = f.sd;
        const bool big = cfg.kind == 1;
        conv_pre = ws.pack_conv(sd, "conv_pre.weight", "conv_pre.bias", 1, 7);
        int ch = cfg.upsample_initial_channel;
        for (int i = 0; i < cfg.n_upsamples; ++i) {
            const std::string name = big ? "ups." + std::to_string(i) + ".0" : "ups." + std::to_string(i);
            UpW u;
            u.stride = cfg.upsample_rates[i];
            u.k = cfg.upsample_kernel_sizes[i];
            u.cin = ch;
            u.cout = ch / 2;
            const int pad = (u.k - u.stride) / 2;
            for (int carry = 0; carry < 2; ++carry) {
                bool any = false;
                for (int r = 0; r < u.stride; ++r) any = any || ((r + pad) / u.stride == carry);
                if (!any) continue;
                u.ph[u.n_groups] = ws.pack_convtr_phase(sd, name + ".weight", name + ".bias", u.stride, pad, carry,
                                                        &u.r_start[u.n_groups], &u.r_count[u.n_groups]);
                // remember the carry in r_start's sign-free companion: recomputed at run time from r_start
                ++u.n_groups;
            }
            ups.push_back(u);
            ch /= 2;
            hop *= u.stride;
            for (int j = 0; j < cfg.n_kernels; ++j) {
                const std::string p = "resblocks." + std::to_string(i * cfg.n_kernels + j) + ".";
                ResBlockW rb;
                for (int m = 0; m < cfg.n_dilations; ++m) {
                    ConvK a, b;
                    a.k = b.k = cfg.resblock_kernel_sizes[j];
                    a.dil = cfg.resblock_dilation_sizes[j][m];
                    b.dil = 1;
                    if (cfg.resblock == 2) {          // hifigan.py:74-80 / bigvgan models.py:95-117: `convs`, `activations`
                        a.w = ws.pack_conv(sd, p + "convs." + std::to_string(m) + ".weight",
                                           p + "convs." + std::to_string(m) + ".bias", 1, a.k);
                        rb.c1.push_back(a);
                        if (big) {
                            float *al, *ib;
                            snake_params(sd, p + "activations." + std::to_string(m) + ".", &al, &ib);
                            rb.alpha.push_back(al);
                            rb.inv_beta.push_back(ib);
                        }
                        continue;
                    }
                    a.w = ws.pack_conv(sd, p + "convs1." + std::to_string(m) + ".weight",
                                       p + "convs1." + std::to_string(m) + ".bias", 1, a.k);
                    b.w = ws.pack_conv(sd, p + "convs2." + std::to_string(m) + ".weight",
                                       p + "convs2." + std::to_string(m) + ".bias", 1, b.k);
                    rb.c1.push_back(a);
                    rb.c2.push_back(b);
                    if (big) {
                        for (int q = 0; q < 2; ++q) {
                            float *al, *ib;
                            snake_params(sd, p + "activations." + std::to_string(2 * m + q) + ".", &al, &ib);
                            rb.alpha.push_back(al);
                            rb.inv_beta.push_back(ib);
                        }
                    }
                }
                rbs.push_back(rb);
            }
        }
        if (big) snake_params(sd, "activation_post.", &post_alpha, &post_inv_beta);
        conv_post = ws.pack_conv(sd, "conv_post.weight", "conv_post.bias", 1, 7);
        if (cfg.use_pitch_embed) {
            MAA_CHECK(!big, "the NSF branch belongs to the HiFi-GAN generator");
            MAA_CHECK(cfg.harmonic_num > 0 && cfg.sampling_rate > 0, "NSF branch needs harmonic_num and sampling_rate");
            const HostTensor& lw = get(sd, "m_source.l_linear.weight");
            MAA_CHECK(lw.numel() == cfg.harmonic_num + 1, "m_source.l_linear.weight shape");
            src_w = ws.vec(sd, "m_source.l_linear.weight");
            src_b = ws.vec(sd, "m_source.l_linear.bias");
            for (int i = 0; i < cfg.n_upsamples; ++i) {
                int s = 1;
                for (int q = i + 1; q < cfg.n_upsamples; ++q) s *= cfg.upsample_rates[q];
                const bool last = i + 1 == cfg.n_upsamples;
                const std::string name = "noise_convs." + std::to_string(i);
                noise_convs.push_back(ws.pack_conv(sd, name + ".weight", name + ".bias", 1, last ? 1 : 2 * s));
                noise_stride.push_back(last ? 1 : s);
            }
        }
    }

    // dilated "same" conv1d on [B, L, C]
    void conv1d(Ctx& ctx, const T4& x, const ConvK& c, float leaky, const float* res, float out_scale, int accumulate,
                T4& out) {
        // narrow stages (C = 32 / 64): input tile staged once in LDS for all taps -- HBM-bound instead of L2-re-read-bound
        if (x.C == out.C && !x.split && x.H == 1 && x.ld == 0 &&
            launch_halo_conv1d(ctx, x.p, x.B, x.W, x.C, c.w, c.k, c.dil, leaky != 0.f ? leaky : 1.f, res, out_scale, accumulate,
                               out.p))
            return;
        ConvOpt o;
        o.KH = 1;
        o.KW = c.k;
        o.dil = c.dil;
        o.pad = (c.k * c.dil - c.dil) / 2;
        o.pad_h = 0;
        if (leaky != 0.f) {
            o.a_act = 1;
            o.a_slope = leaky;
        }
        o.res = res;
        o.out_scale = out_scale;
        o.accumulate = accumulate;
        conv_into(ctx, x, nullptr, c.w, o, out);
    }

    // The same conv on a PRE-ACTIVATED, PRE-SPLIT input (split32 lines of leaky(x)): both operands go to LDS by DMA and the
    // contraction runs on the ping-pong engine (igemm_pp.hip).  post_slope != 0: the output is leaky(result) as split32 lines
    // (c1 of an MRF pair: only the next convolution reads it); else fp32 with the residual / scale / accumulate epilogue, and
    // next_split != null additionally receives leaky(result) as split32 lines -- the next pair's input.
    void conv1d_split(Ctx& ctx, const T4& xs, const ConvK& c, float post_slope, const float* res, float out_scale,
                      int accumulate, T4& out, float* next_split, float next_slope) {
        ConvOpt o;
        o.KH = 1;
        o.KW = c.k;
        o.dil = c.dil;
        o.pad = (c.k * c.dil - c.dil) / 2;
        o.pad_h = 0;
        o.res = res;
        o.out_scale = out_scale;
        o.accumulate = accumulate;
        if (post_slope != 0.f) {
            o.act = 4;
            o.act_slope = post_slope;
            o.c_split = 1;
        }
        o.c2 = next_split;
        o.c2_slope = next_slope;
        conv_into(ctx, xs, nullptr, c.w, o, out);
    }

    // har: harmonic source [B, T*hop] of the NSF branch, or null
    void forward(Ctx& ctx, const float* mel, int B, int T, float* wav, const float* har = nullptr) {
        const bool big = cfg.kind == 1;
        T4 m = alloc_t(ctx, B, 1, T, cfg.num_mels);
        launch_nchw_to_nhwc(ctx, mel, B, cfg.num_mels, T, m.p);
        T4 x = alloc_t(ctx, B, 1, T, cfg.upsample_initial_channel);
        {
            ConvOpt o;
            o.KW = 7;
            o.pad = 3;
            o.pad_h = 0;
            conv_into(ctx, m, nullptr, conv_pre, o, x);
        }
        int L = T;
        for (size_t i = 0; i < ups.size(); ++i) {
            const UpW& u = ups[i];
            const int Lo = L * u.stride;
            T4 y = alloc_t(ctx, B, 1, Lo, u.cout);
            const int U = u.k / u.stride, pad = (u.k - u.stride) / 2;
            for (int gi = 0; gi < u.n_groups; ++gi) {
                // polyphase group: rows j of [B*L] produce output rows s*j + r, r in [r_start, r_start + r_count)
                const int carry = (u.r_start[gi] + pad) / u.stride;
                IGemm p;
                p.a1 = x.p;
                p.lda1 = u.cin;
                p.C1 = u.cin;
                p.Hin = 1;
                p.Win = L;
                p.Hout = 1;
                p.Wout = L;
                p.KH = 1;
                p.KW = U;
                p.pw = U - 1 - carry;
                if (!big) {                       // hifigan.py:153 leaky before ups; BigVGAN has none (models.py:185-188)
                    p.a_act = 1;
                    p.a_slope = 0.1f;
                }
                p.b = u.ph[gi].w;
                p.ldb = u.ph[gi].ld;
                p.b_nk = u.ph[gi].nk;
                p.b_split = u.ph[gi].split;
                p.M = B * L;
                p.K = U * u.cin;
                p.N = u.r_count[gi] * u.cout;
                p.bias = u.ph[gi].bias;
                p.c = y.p + (long long)u.r_start[gi] * u.cout;
                p.ldc = u.stride * u.cout;
                launch_igemm(ctx, p);
            }
            L = Lo;
            if (har) {      // x = x + noise_convs[i](har_source)   (hifigan.py:155-157; Conv1d(1, C, 2s, stride s, pad s/2))
                T4 hs;
                hs.B = B;
                hs.H = 1;
                hs.W = T * hop;
                hs.C = 1;
                hs.p = const_cast<float*>(har);
                ConvOpt o;
                o.KH = 1;
                o.KW = noise_convs[i].K;
                o.stride = noise_stride[i];
                o.pad = noise_stride[i] > 1 ? noise_stride[i] / 2 : 0;
                o.pad_h = 0;
                o.accumulate = 1;
                MAA_CHECK((hs.W + 2 * o.pad - o.KW) / o.stride + 1 == L, "noise_convs output length");
                conv_into(ctx, hs, nullptr, noise_convs[i], o, y);
            }
            // MRF: x = (rb_0(y) + rb_1(y) + rb_2(y)) / n on the SAME input (hifigan.py:158-164)
            T4 xs = alloc_t(ctx, B, 1, L, u.cout);
            const float inv_n = 1.0f / (float)cfg.n_kernels;
            // Wide HiFi-GAN stages in the bf16 modes: every MRF convolution reads a pre-activated, pre-split input --
            // leaky(x) as split32 lines, written by the producer's epilogue -- so that both operands reach LDS by DMA and
            // the contraction runs on the ping-pong engine (C >= 128: the layer is matrix work, not HBM traffic; the
            // narrow stages keep the halo kernel).
            const bool presplit = !big && ctx.dtype != 0 && u.cout >= 128 && u.cout % 32 == 0;
            T4 ys;
            if (presplit) {
                ys = alloc_t(ctx, B, 1, L, u.cout);
                ys.split = true;
                launch_split32_pack(ctx, y.p, (long long)B * L, u.cout, ys.p, 0.1f);
            }
            const size_t mk = ctx.ws.mark();      // (per-resblock scratch below is given back after each resblock; ys stays)
            for (int j = 0; j < cfg.n_kernels; ++j) {
                const ResBlockW& rb = rbs[i * cfg.n_kernels + j];
                T4 cur = y;
                T4 t1 = alloc_t(ctx, B, 1, L, u.cout);
                T4 bufA = alloc_t(ctx, B, 1, L, u.cout), bufB = alloc_t(ctx, B, 1, L, u.cout);
                T4 act = big ? alloc_t(ctx, B, 1, L, u.cout) : T4();
                T4 sA, sB, cur_s = ys;
                if (presplit) {
                    sA = alloc_t(ctx, B, 1, L, u.cout);
                    sB = alloc_t(ctx, B, 1, L, u.cout);
                    sA.split = sB.split = t1.split = true;
                }
                for (size_t mth = 0; mth < rb.c1.size(); ++mth) {
                    const bool last = mth + 1 == rb.c1.size();
                    if (cfg.resblock == 2) {
                        // xt = c(act(x)); x = xt + x                      (hifigan.py:83-88 / bigvgan models.py:122-128)
                        if (presplit) {
                            if (last) {
                                conv1d_split(ctx, cur_s, rb.c1[mth], 0.f, cur.p, inv_n, j > 0, xs, nullptr, 1.f);
                            } else {
                                T4& dst = (cur.p == bufA.p) ? bufB : bufA;
                                T4& dst_s = (cur_s.p == sA.p) ? sB : sA;
                                conv1d_split(ctx, cur_s, rb.c1[mth], 0.f, cur.p, 1.f, 0, dst, dst_s.p, 0.1f);
                                cur = dst;
                                cur_s = dst_s;
                            }
                            continue;
                        }
                        if (big) launch_snake_aa(ctx, cur.p, B, L, u.cout, rb.inv_beta[mth], rb.alpha[mth], act.p);
                        const T4& in1 = big ? act : cur;
                        const float lk1 = big ? 0.f : 0.1f;
                        if (last) {
                            conv1d(ctx, in1, rb.c1[mth], lk1, cur.p, inv_n, j > 0, xs);
                        } else {
                            T4& dst = (cur.p == bufA.p) ? bufB : bufA;
                            conv1d(ctx, in1, rb.c1[mth], lk1, cur.p, 1.f, 0, dst);
                            cur = dst;
                        }
                        continue;
                    }
                    // xt = c1(act(x)); xt = c2(act(xt)); x = xt + x      (hifigan.py:54-61 / bigvgan models.py:72-81)
                    if (presplit) {
                        conv1d_split(ctx, cur_s, rb.c1[mth], 0.1f, nullptr, 1.f, 0, t1, nullptr, 1.f);      // t1 = split(leaky(c1(.)))
                        if (last) {
                            conv1d_split(ctx, t1, rb.c2[mth], 0.f, cur.p, inv_n, j > 0, xs, nullptr, 1.f);
                        } else {
                            T4& dst = (cur.p == bufA.p) ? bufB : bufA;
                            T4& dst_s = (cur_s.p == sA.p) ? sB : sA;
                            conv1d_split(ctx, t1, rb.c2[mth], 0.f, cur.p, 1.f, 0, dst, dst_s.p, 0.1f);
                            cur = dst;
                            cur_s = dst_s;
                        }
                        continue;
                    }
                    if (big) {
                        launch_snake_aa(ctx, cur.p, B, L, u.cout, rb.inv_beta[2 * mth], rb.alpha[2 * mth], act.p);
                        conv1d(ctx, act, rb.c1[mth], 0.f, nullptr, 1.f, 0, t1);
                        launch_snake_aa(ctx, t1.p, B, L, u.cout, rb.inv_beta[2 * mth + 1], rb.alpha[2 * mth + 1], act.p);
                    } else {
                        // narrow stages (C = 32 / 64): the pair in one launch, xt stays in LDS (halo_conv1d.hip)
                        T4& dstp = last ? xs : ((cur.p == bufA.p) ? bufB : bufA);
                        if (!cur.split && cur.ld == 0 &&
                            launch_halo_pair(ctx, cur.p, B, L, u.cout, rb.c1[mth].w, rb.c1[mth].k, rb.c1[mth].dil, 0.1f, rb.c2[mth].w,
                                             rb.c2[mth].k, rb.c2[mth].dil, 0.1f, cur.p, last ? inv_n : 1.f, last ? (j > 0) : 0, dstp.p)) {
                            if (!last) cur = dstp;
                            continue;
                        }
                        conv1d(ctx, cur, rb.c1[mth], 0.1f, nullptr, 1.f, 0, t1);
                    }
                    const T4& in2 = big ? act : t1;
                    const float lk2 = big ? 0.f : 0.1f;
                    if (last) {
                        conv1d(ctx, in2, rb.c2[mth], lk2, cur.p, inv_n, j > 0, xs);
                    } else {
                        T4& dst = (cur.p == bufA.p) ? bufB : bufA;
                        conv1d(ctx, in2, rb.c2[mth], lk2, cur.p, 1.f, 0, dst);
                        cur = dst;
                    }
                }
                ctx.ws.release(mk);
            }
            x = xs;
        }
        // final activation (leaky slope 0.01, hifigan.py:165 / activation_post, bigvgan models.py:199), conv_post, tanh
        T4 w;            // [B, L, 1] is the caller's [B, L] wave buffer
        w.B = B;
        w.H = 1;
        w.W = L;
        w.C = 1;
        w.p = wav;
        ConvOpt o;
        o.KW = 7;
        o.pad = 3;
        o.pad_h = 0;
        o.act = 1;
        if (big) {
            T4 act = alloc_t(ctx, B, 1, L, x.C);
            launch_snake_aa(ctx, x.p, B, L, x.C, post_inv_beta, post_alpha, act.p);
            conv_into(ctx, act, nullptr, conv_post, o, w);
        } else {
            o.a_act = 1;
            o.a_slope = 0.01f;
            conv_into(ctx, x, nullptr, conv_post, o, w);
        }
    }
};

Vocoder::Vocoder(const maa_vocoder_config& cfg, const StateDict& sd, int precision) : impl_(new Impl(precision)) {
    MAA_CHECK(cfg.resblock >= 0 && cfg.resblock <= 2, "maa_vocoder_config.resblock is 1 or 2 (0 = 1)");
    impl_->cfg = cfg;
    impl_->build(sd);
}
Vocoder::~Vocoder() { delete impl_; }
int Vocoder::hop() const { return impl_->hop; }

void Vocoder::forward(Ctx& ctx, const float* mel, int B, int T, float* wav) {
    // a generator built with use_pitch_embed called without f0 simply skips the source branch, as HifiGanGenerator.forward(x,
    // f0=None) does (hifigan.py:144-169: `if self.use_pitch_embed and f0 is not None`)
    PrecisionGuard pg(ctx, impl_->precision);
    run_sized(ctx, [&] { impl_->forward(ctx, mel, B, T, wav); });
}

void Vocoder::forward_f0(Ctx& ctx, const float* mel, const float* f0, const float* rand_ini, const float* noise, int B,
                         int T, float* wav) {
    Impl& m = *impl_;
    MAA_CHECK(m.cfg.use_pitch_embed, "generator built without use_pitch_embed has no NSF branch");
    PrecisionGuard pg(ctx, m.precision);
    run_sized(ctx, [&] {
        const long long n = (long long)B * T * m.hop;
        float* har = ctx.ws.alloc_f((size_t)n);
        const size_t mk = ctx.ws.mark();
        float* sines = ctx.ws.alloc_f((size_t)n * (m.cfg.harmonic_num + 1));
        launch_nsf_source(ctx, f0, B, T, m.hop, (float)m.cfg.sampling_rate, rand_ini, noise, m.cfg.harmonic_num, m.src_w,
                          m.src_b, sines, har);
        ctx.ws.release(mk);
        m.forward(ctx, mel, B, T, wav, har);
    });
}

}  // namespace maa
