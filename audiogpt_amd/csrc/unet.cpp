// Latent-diffusion UNet executor (eps-prediction) on channels-last fp32 activations.
//
// Mirrors the graph of (paths relative to text_to_audio/Make_An_Audio in the reference):
//   ldm/modules/diffusionmodules/openaimodel.py:516-744 (ctor + forward), :255-275 ResBlock,
//   :317-324,356-372 AttentionBlock/QKVAttentionLegacy (inpaint), :134-160 Downsample, :91-119 Upsample
//   ldm/modules/diffusionmodules/custom_openaimodel.py:352-354 (I2A: emb += context.squeeze(1))
//   ldm/modules/attention.py:37-64,152-261 (GEGLU FF, CrossAttention, BasicTransformerBlock, SpatialTransformer)
// Differences from a literal translation (same maths, fewer passes over HBM):
//   * no torch.cat for the skip connections: GroupNorm and the 1x1 skip conv read (h | skip) as two sources
//   * all ResBlock `emb_layers` Linear(SiLU(emb)) run as ONE GEMM per forward, sliced per block
//   * to_q/to_k/to_v of self-attention are one GEMM (N = 3*inner); the cross-attention K/V projections of the
//     (step-invariant) context are computed once per `set_context` and reused by every DDIM step
//   * bias, time-embedding add, residual add and GEGLU are igemm epilogues
//   * a guided DDIM step evaluates the model on cat([x] * 2) with cat([uncond, cond]) (ddim.py:177-199): the two halves differ
//     only in the context, so every layer before the first cross-attention (conv_in, the first ResBlock, the first
//     transformer's norm / proj_in / self-attention / to_q) is the same tensor twice -- computed on one half and duplicated
//     where the halves part (forward_body `dup`; round 6).  Every kernel is batch-invariant, so the result is bit-identical.
#include "models.h"

#include <atomic>
#include <cmath>
#include <cstdlib>

namespace maa {

namespace {

struct ResW {
    int cin = 0, cout = 0, updown = 0;   // 0 none, 1 down (avg-pool), 2 up (nearest)
    float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
    PackedW conv1, conv2, skip;
    bool has_skip = false;
    int emb_off = 0;
};
struct STBlockW {
    float *ln1g, *ln1b, *ln2g, *ln2b, *ln3g, *ln3b;
    PackedW qkv1, out1, q2, kv2, out2, ff1, ff2;
    int kv_slot = -1;
};
struct STW {
    int ch = 0, heads = 0, dh = 0;
    float *ng = nullptr, *nb = nullptr;
    PackedW proj_in, proj_out;
    std::vector<STBlockW> blocks;
};
struct AttnW {
    int ch = 0, heads = 0;
    float *ng = nullptr, *nb = nullptr;
    PackedW qkv, proj;
};
struct ConvW {
    PackedW w, w_up2;      // w_up2: the four 2x2 phase weights of an Upsample convolution (pack_conv_up2), else empty
    int cin = 0, cout = 0;
};
enum Kind { kConv, kRes, kST, kAttn, kDown, kUp };
struct Layer {
    Kind kind;
    int idx;
};

}  // namespace

struct UNet::Impl {
    maa_unet_config cfg;
    int precision = 0;
    WeightStore ws;
    explicit Impl(int prec) : precision(prec), ws(prec != 0) {}
    std::vector<ResW> res;
    std::vector<STW> st;
    std::vector<AttnW> attn;
    std::vector<ConvW> convs;
    std::vector<std::vector<Layer>> input, output;
    std::vector<Layer> middle;
    PackedW te0, te2, emb_all, out_conv, out_narrow;
    float *out_g = nullptr, *out_b = nullptr;
    int emb_dim = 0, emb_total = 0;
    // cross-attention K/V cache: one [rows, 2*inner] buffer per transformer block
    unsigned long long serial = next_serial();
    static unsigned long long next_serial() {
        static std::atomic<unsigned long long> n{1};
        return n.fetch_add(1);
    }
    std::vector<float*> kv_cache;
    std::vector<int> kv_inner;
    int kv_rows = 0, kv_len = 0, kv_batch = 0;
    size_t kv_cap_rows = 0;
    DevSlab cfg_context;      // [uncond ; cond] rows of a CFG sample() call
    // The shared prefix of a guided step run as two lanes (ddim.cpp): the unconditional lane computes the layers before the
    // first cross-attention as part of its own pass and leaves their four outputs where they are in ITS workspace (share 2);
    // the conditional lane starts at that cross-attention and reads them (share 3) -- `ready` orders the two streams, the
    // step's join orders the next step's writes behind the reads.
    struct Prefix {
        const float *hs0 = nullptr, *xres = nullptr, *y1 = nullptr, *q = nullptr;
        int B = 0, H = 0, W = 0;
        hipEvent_t ready = nullptr;
    } prefix;
    int share = 0;            // this forward: 0 whole network, 1 one stream [x ; x] (dup), 2 lane that exports, 3 lane that imports

    // conv_in, then [ResBlock, SpatialTransformer]: the shape the lanes' hand-over is written for (T2A; anything else keeps
    // the whole evaluation per lane)
    bool prefix_ok() const {
        return cfg.use_spatial_transformer && !cfg.add_context_to_emb && input.size() >= 2 && input[0].size() == 1 &&
               input[0][0].kind == kConv && input[1].size() == 2 && input[1][0].kind == kRes && input[1][1].kind == kST &&
               !st[input[1][1].idx].blocks.empty();
    }

    ~Impl() {
        for (float* p : kv_cache)
            if (p) (void)hipFree(p);
        if (prefix.ready) (void)hipEventDestroy(prefix.ready);
    }

    // ---- construction ---------------------------------------------------------------------------
    Layer add_res(const StateDict& sd, const std::string& p, int cin, int cout, int updown) {
        ResW r;
        r.cin = cin;
        r.cout = cout;
        r.updown = updown;
        r.g1 = ws.vec(sd, p + "in_layers.0.weight");
        r.b1 = ws.vec(sd, p + "in_layers.0.bias");
        r.conv1 = ws.pack_conv(sd, p + "in_layers.2.weight", p + "in_layers.2.bias", 3, 3);
        r.g2 = ws.vec(sd, p + "out_layers.0.weight");
        r.b2 = ws.vec(sd, p + "out_layers.0.bias");
        r.conv2 = ws.pack_conv(sd, p + "out_layers.3.weight", p + "out_layers.3.bias", 3, 3);
        r.has_skip = has(sd, p + "skip_connection.weight");
        if (r.has_skip) r.skip = ws.pack_conv(sd, p + "skip_connection.weight", p + "skip_connection.bias", 1, 1);
        MAA_CHECK(r.has_skip == (cin != cout), "skip_connection presence " + p);
        r.emb_off = emb_total;
        emb_total += cout;
        emb_w.push_back(p + "emb_layers.1.weight");
        emb_b.push_back(p + "emb_layers.1.bias");
        res.push_back(r);
        return {kRes, (int)res.size() - 1};
    }
    Layer add_st(const StateDict& sd, const std::string& p, int ch, int heads, int dh) {
        STW s;
        s.ch = ch;
        s.heads = heads;
        s.dh = dh;
        const int inner = heads * dh;
        s.ng = ws.vec(sd, p + "norm.weight");
        s.nb = ws.vec(sd, p + "norm.bias");
        s.proj_in = ws.pack_conv(sd, p + "proj_in.weight", p + "proj_in.bias", 1, 1);
        s.proj_out = ws.pack_conv(sd, p + "proj_out.weight", p + "proj_out.bias", 1, 1);
        for (int d = 0; d < cfg.transformer_depth; ++d) {
            const std::string t = p + "transformer_blocks." + std::to_string(d) + ".";
            STBlockW b;
            b.ln1g = ws.vec(sd, t + "norm1.weight");
            b.ln1b = ws.vec(sd, t + "norm1.bias");
            b.ln2g = ws.vec(sd, t + "norm2.weight");
            b.ln2b = ws.vec(sd, t + "norm2.bias");
            b.ln3g = ws.vec(sd, t + "norm3.weight");
            b.ln3b = ws.vec(sd, t + "norm3.bias");
            b.qkv1 = ws.pack_concat(sd, {t + "attn1.to_q.weight", t + "attn1.to_k.weight", t + "attn1.to_v.weight"}, {});
            b.out1 = ws.pack_conv(sd, t + "attn1.to_out.0.weight", t + "attn1.to_out.0.bias", 1, 1);
            b.q2 = ws.pack_conv(sd, t + "attn2.to_q.weight", "", 1, 1);
            b.kv2 = ws.pack_concat(sd, {t + "attn2.to_k.weight", t + "attn2.to_v.weight"}, {});
            b.out2 = ws.pack_conv(sd, t + "attn2.to_out.0.weight", t + "attn2.to_out.0.bias", 1, 1);
            b.ff1 = ws.pack_geglu(sd, t + "ff.net.0.proj.weight", t + "ff.net.0.proj.bias");
            b.ff2 = ws.pack_conv(sd, t + "ff.net.2.weight", t + "ff.net.2.bias", 1, 1);
            b.kv_slot = (int)kv_cache.size();
            kv_cache.push_back(nullptr);
            kv_inner.push_back(inner);
            s.blocks.push_back(b);
        }
        st.push_back(s);
        return {kST, (int)st.size() - 1};
    }
    Layer add_attn(const StateDict& sd, const std::string& p, int ch, int heads) {
        AttnW a;
        a.ch = ch;
        a.heads = heads;
        a.ng = ws.vec(sd, p + "norm.weight");
        a.nb = ws.vec(sd, p + "norm.bias");
        a.qkv = ws.pack_conv(sd, p + "qkv.weight", p + "qkv.bias", 1, 1);
        a.proj = ws.pack_conv(sd, p + "proj_out.weight", p + "proj_out.bias", 1, 1);
        attn.push_back(a);
        return {kAttn, (int)attn.size() - 1};
    }
    Layer add_conv(const StateDict& sd, const std::string& p, int cin, int cout, Kind kind) {
        ConvW c;
        c.cin = cin;
        c.cout = cout;
        c.w = ws.pack_conv(sd, p + "weight", p + "bias", 3, 3);
        if (kind == kUp) c.w_up2 = ws.pack_conv_up2(sd, p + "weight", p + "bias");
        convs.push_back(c);
        return {kind, (int)convs.size() - 1};
    }
    std::vector<std::string> emb_w, emb_b;

    int cur_heads = 0;
    Layer attn_layer(const StateDict& sd, const std::string& p, int ch) {
        // openaimodel.py:534-553 head bookkeeping
        int heads, dh;
        if (cfg.num_head_channels == -1) {
            heads = cur_heads;
            dh = ch / heads;
        } else {
            heads = ch / cfg.num_head_channels;
            cur_heads = heads;
            dh = cfg.num_head_channels;
        }
        if (cfg.legacy) dh = cfg.use_spatial_transformer ? ch / heads : cfg.num_head_channels;
        if (cfg.use_spatial_transformer) return add_st(sd, p, ch, heads, dh);
        return add_attn(sd, p, ch, dh == -1 ? heads : ch / dh);
    }

    void build(const StateDict& sd) {
        const int mc = cfg.model_channels;
        emb_dim = mc * 4;
        cur_heads = cfg.num_heads;
        te0 = ws.pack_conv(sd, "time_embed.0.weight", "time_embed.0.bias", 1, 1);
        te2 = ws.pack_conv(sd, "time_embed.2.weight", "time_embed.2.bias", 1, 1);
        auto in_attn = [&](int ds) {
            for (int i = 0; i < cfg.n_attention_resolutions; ++i)
                if (cfg.attention_resolutions[i] == ds) return true;
            return false;
        };
        std::vector<int> chans;
        input.push_back({add_conv(sd, "input_blocks.0.0.", cfg.in_channels, mc, kConv)});
        chans.push_back(mc);
        int ch = mc, ds = 1;
        for (int level = 0; level < cfg.n_channel_mult; ++level) {
            const int m = cfg.channel_mult[level];
            for (int i = 0; i < cfg.num_res_blocks; ++i) {
                const std::string p = "input_blocks." + std::to_string(input.size()) + ".";
                std::vector<Layer> blk;
                blk.push_back(add_res(sd, p + "0.", ch, m * mc, 0));
                ch = m * mc;
                if (in_attn(ds)) blk.push_back(attn_layer(sd, p + "1.", ch));
                input.push_back(blk);
                chans.push_back(ch);
            }
            if (level != cfg.n_channel_mult - 1) {
                const std::string p = "input_blocks." + std::to_string(input.size()) + ".0.";
                if (cfg.resblock_updown)
                    input.push_back({add_res(sd, p, ch, ch, 1)});
                else
                    input.push_back({add_conv(sd, p + "op.", ch, ch, kDown)});
                chans.push_back(ch);
                ds *= 2;
            }
        }
        middle.push_back(add_res(sd, "middle_block.0.", ch, ch, 0));
        middle.push_back(attn_layer(sd, "middle_block.1.", ch));
        middle.push_back(add_res(sd, "middle_block.2.", ch, ch, 0));
        for (int level = cfg.n_channel_mult - 1; level >= 0; --level) {
            const int m = cfg.channel_mult[level];
            for (int i = 0; i < cfg.num_res_blocks + 1; ++i) {
                const int ich = chans.back();
                chans.pop_back();
                const std::string p = "output_blocks." + std::to_string(output.size()) + ".";
                std::vector<Layer> blk;
                blk.push_back(add_res(sd, p + "0.", ch + ich, mc * m, 0));
                ch = mc * m;
                int j = 1;
                if (in_attn(ds)) blk.push_back(attn_layer(sd, p + std::to_string(j++) + ".", ch));
                if (level && i == cfg.num_res_blocks) {
                    const std::string q = p + std::to_string(j++) + ".";
                    if (cfg.resblock_updown)
                        blk.push_back(add_res(sd, q, ch, ch, 2));
                    else
                        blk.push_back(add_conv(sd, q + "conv.", ch, ch, kUp));
                    ds /= 2;
                }
                output.push_back(blk);
            }
        }
        out_g = ws.vec(sd, "out.0.weight");
        out_b = ws.vec(sd, "out.0.bias");
        out_conv = ws.pack_conv(sd, "out.2.weight", "out.2.bias", 3, 3);
        out_narrow = ws.pack_narrow3x3(sd, "out.2.weight", "out.2.bias");
        emb_all = ws.pack_concat(sd, emb_w, emb_b);
        MAA_CHECK(emb_all.N == emb_total, "emb_layers packing");
    }

    // ---- execution ------------------------------------------------------------------------------
    T4 run_res(Ctx& ctx, const ResW& r, const T4& x1, const T4* x2, const float* emb_out) {
        const int B = x1.B, H = x1.H, W = x1.W;
        const int Ho = r.updown == 1 ? H / 2 : (r.updown == 2 ? H * 2 : H);
        const int Wo = r.updown == 1 ? W / 2 : (r.updown == 2 ? W * 2 : W);
        T4 out = alloc_t(ctx, B, Ho, Wo, r.cout);
        const size_t mk = ctx.ws.mark();
        T4 t1 = alloc_t(ctx, B, H, W, r.cin);
        t1.split = r.updown != 1 && split_for_gemm(ctx, r.cin);     // (the avg-pool of a down block reads fp32)
        // a ResBlock that changes the channel count runs a 1x1 skip convolution over the same (h | skip) rows GroupNorm reads:
        // the apply pass also writes them as split32, so that contraction takes the LDS-DMA engine (both operands by DMA; same
        // products in the same order as the register-staged engine it used to run on -- bit-identical)
        T4 xraw;
        if (r.has_skip && r.updown == 0 && split_for_gemm(ctx, r.cin) && r.skip.split) {
            xraw = alloc_t(ctx, B, H, W, r.cin);
            xraw.split = true;
        }
        launch_groupnorm(ctx, x1.p, x1.C, x1.C, x2 ? x2->p : nullptr, x2 ? x2->C : 0, x2 ? x2->C : 0, B, H * W, 32,
                         r.g1, r.b1, 1e-5f, 1, t1.p, t1.split, xraw.p);
        T4 h1 = alloc_t(ctx, B, Ho, Wo, r.cout);
        ConvOpt o1;
        o1.KH = o1.KW = 3;
        o1.pad = 1;
        o1.rowadd = emb_out + r.emb_off;
        o1.ld_rowadd = emb_ld;
        const float* resid = nullptr;
        if (r.updown == 1) {          // openaimodel.py:212-214,256-261: avg-pool both h and x, then conv
            MAA_CHECK(!x2, "down ResBlock takes one source");
            T4 tp = alloc_t(ctx, B, Ho, Wo, r.cin), xp = alloc_t(ctx, B, Ho, Wo, r.cin);
            launch_avgpool2(ctx, t1.p, B, H, W, r.cin, tp.p);
            launch_avgpool2(ctx, x1.p, B, H, W, r.cin, xp.p);
            conv_into(ctx, tp, nullptr, r.conv1, o1, h1);
            resid = xp.p;
        } else if (r.updown == 2) {   // :209-211: nearest-2x on both; the conv reads t1 through the gather
            MAA_CHECK(!x2, "up ResBlock takes one source");
            T4 xu = alloc_t(ctx, B, Ho, Wo, r.cin);
            launch_upsample2(ctx, x1.p, B, H, W, r.cin, xu.p);
            o1.up = 1;
            conv_into(ctx, t1, nullptr, r.conv1, o1, h1);
            resid = xu.p;
        } else {
            conv_into(ctx, t1, nullptr, r.conv1, o1, h1);
            resid = x1.p;
        }
        T4 t2 = alloc_t(ctx, B, Ho, Wo, r.cout);
        t2.split = split_for_gemm(ctx, r.cout);
        launch_groupnorm(ctx, h1.p, r.cout, r.cout, nullptr, 0, 0, B, Ho * Wo, 32, r.g2, r.b2, 1e-5f, 1, t2.p, t2.split);
        if (r.has_skip) {
            MAA_CHECK(r.updown == 0, "skip conv with up/down");
            T4 sk = alloc_t(ctx, B, H, W, r.cout);
            ConvOpt os;
            if (xraw.p)
                conv_into(ctx, xraw, nullptr, r.skip, os, sk);
            else
                conv_into(ctx, x1, x2, r.skip, os, sk);
            resid = sk.p;
        } else {
            MAA_CHECK(!x2, "identity skip needs a single source");
        }
        ConvOpt o2;
        o2.KH = o2.KW = 3;
        o2.pad = 1;
        o2.res = resid;
        conv_into(ctx, t2, nullptr, r.conv2, o2, out);
        ctx.ws.release(mk);
        return out;
    }

    // expand: x holds ONE half of a guided step's batch (both halves are equal up to here); the block's first cross-attention
    // is where they part, so everything before it runs on x.B samples and the output has 2 x.B
    // xp: 2 = this lane leaves y1 / q of the first block for the other lane (allocated outside the block's scratch scope so that
    // they outlive it), 3 = this lane starts at the first block's cross-attention with the other lane's y1 / q / block input
    T4 run_st(Ctx& ctx, const STW& s, const T4& x, bool expand = false, int xp = 0) {
        const int HW = x.H * x.W, inner = s.heads * s.dh;
        int B = x.B;                                   // samples of the current tensors
        const int Bo = expand ? 2 * x.B : x.B;
        long long M = (long long)B * HW;
        const long long Mo = (long long)Bo * HW;
        T4 out = alloc_t(ctx, Bo, x.H, x.W, s.ch);
        float *y1_keep = nullptr, *q_keep = nullptr;
        if (xp == 2) {
            y1_keep = ctx.ws.alloc_f((size_t)M * inner);
            q_keep = ctx.ws.alloc_f((size_t)M * inner);
        }
        const size_t mk = ctx.ws.mark();
        const float* xres = x.p;                       // proj_out's residual: the block input, for every output sample
        if (expand) {
            float* xd = ctx.ws.alloc_f((size_t)Mo * s.ch);
            launch_dup_half(ctx, x.p, M * s.ch, xd);
            xres = xd;
        }
        const bool sp_in = split_for_gemm(ctx, s.ch), sp = split_for_gemm(ctx, inner);
        float* y = nullptr;
        if (xp != 3) {
            float* xn = ctx.ws.alloc_f((size_t)M * s.ch);
            launch_groupnorm(ctx, x.p, s.ch, s.ch, nullptr, 0, 0, B, HW, 32, s.ng, s.nb, 1e-6f, 0, xn, sp_in);
            y = ctx.ws.alloc_f((size_t)M * inner);
            linear_into(ctx, xn, s.ch, M, s.ch, s.proj_in, nullptr, 0, y, inner, 0, 0, sp_in ? M : 0);
        }
        const float scale = 1.0f / std::sqrt((float)s.dh);
        // attention outputs and the GEGLU product feed exactly one projection each: in the bf16 modes they are written
        // as split32 rows, so to_out / ff.net.2 take the LDS-DMA engine with no per-tile conversion
        const bool no_osplit = false;
        const int o_sp = !no_osplit && sp && flash_attention_covers(ctx, s.dh) ? 1 : 0;
        const int g_sp = !no_osplit && split_for_gemm(ctx, 4 * inner) ? 1 : 0;
        int y_sp = 0;
        for (const STBlockW& b : s.blocks) {
            const bool first_block = &b == &s.blocks.front();
            float* ln = ctx.ws.alloc_f((size_t)Mo * inner);
            float* o = ctx.ws.alloc_f((size_t)Mo * inner);
            float *y1 = nullptr, *q = nullptr;
            if (xp == 3 && first_block) {
                // the other lane computed this far on the same x: its residual stream and queries, read in place
                y1 = const_cast<float*>(prefix.y1);
                q = const_cast<float*>(prefix.q);
            } else {
                // x = attn1(norm1(x)) + x      (attention.py:212)
                launch_layernorm(ctx, y, M, inner, b.ln1g, b.ln1b, 1e-5f, ln, sp);
                float* qkv = ctx.ws.alloc_f((size_t)M * 3 * inner);
                linear_into(ctx, ln, inner, M, inner, b.qkv1, nullptr, 0, qkv, 3 * inner, 0, 0, sp ? M : 0);
                attention_into(ctx, qkv, 3 * inner, s.dh, qkv + inner, 3 * inner, s.dh, qkv + 2 * inner, 3 * inner, s.dh,
                               B, s.heads, s.dh, HW, HW, scale, o, inner, o_sp);
                y1 = xp == 2 && first_block ? y1_keep : ctx.ws.alloc_f((size_t)M * inner);
                linear_into(ctx, o, inner, M, inner, b.out1, y, inner, y1, inner, 0, 0, o_sp ? M : 0);
                // x = attn2(norm2(x), context) + x      (:213)
                launch_layernorm(ctx, y1, M, inner, b.ln2g, b.ln2b, 1e-5f, ln, sp);
                q = xp == 2 && first_block ? q_keep : ctx.ws.alloc_f((size_t)M * inner);
                linear_into(ctx, ln, inner, M, inner, b.q2, nullptr, 0, q, inner, 0, 0, sp ? M : 0);
                if (xp == 2 && first_block) {
                    prefix.y1 = y1;
                    prefix.q = q;
                    prefix.xres = x.p;
                    if (!ctx.ws.dry) MAA_HIP(hipEventRecord(prefix.ready, ctx.stream));
                }
            }
            if (B != Bo) {      // the halves part here: the same queries and the same residual stream meet two contexts
                float* qd = ctx.ws.alloc_f((size_t)Mo * inner);
                float* yd = ctx.ws.alloc_f((size_t)Mo * inner);
                launch_dup_half(ctx, q, M * inner, qd);
                launch_dup_half(ctx, y1, M * inner, yd);
                q = qd;
                y1 = yd;
                B = Bo;
                M = Mo;
            }
            MAA_CHECK(ctx.ws.dry || (kv_cache[b.kv_slot] && (lane ? batch_off + B <= kv_batch : kv_batch == B)),
                      "set_context must precede forward (batch)");
            const float* kv = kv_cache[b.kv_slot] + (size_t)batch_off * kv_len * 2 * inner;
            attention_into(ctx, q, inner, s.dh, kv, 2 * inner, s.dh, kv + inner, 2 * inner, s.dh, B, s.heads, s.dh, HW,
                           kv_len, scale, o, inner, o_sp);
            float* y2 = ctx.ws.alloc_f((size_t)M * inner);
            linear_into(ctx, o, inner, M, inner, b.out2, y1, inner, y2, inner, 0, 0, o_sp ? M : 0);
            // x = ff(norm3(x)) + x      (:214)
            launch_layernorm(ctx, y2, M, inner, b.ln3g, b.ln3b, 1e-5f, ln, sp);
            float* g = ctx.ws.alloc_f((size_t)M * 4 * inner);
            linear_into(ctx, ln, inner, M, inner, b.ff1, nullptr, 0, g, 4 * inner, /*geglu=*/1, 0, sp ? M : 0, g_sp);
            float* y3 = ctx.ws.alloc_f((size_t)M * inner);
            // the last block's output feeds only proj_out: written as split32 as well (bias + residual are applied
            // before the split, in the same epilogue)
            const bool last = &b == &s.blocks.back();
            y_sp = last && sp && !no_osplit ? 1 : 0;
            linear_into(ctx, g, 4 * inner, M, 4 * inner, b.ff2, y2, inner, y3, inner, 0, 0, g_sp ? M : 0, y_sp);
            y = y3;
        }
        MAA_CHECK(B == Bo, "SpatialTransformer without a transformer block cannot part the halves of a guided step");
        linear_into(ctx, y, inner, M, inner, s.proj_out, xres, s.ch, out.p, s.ch, 0, 0, y_sp ? M : 0);
        ctx.ws.release(mk);
        return out;
    }

    T4 run_attn(Ctx& ctx, const AttnW& a, const T4& x) {
        // openaimodel.py:317-324 + QKVAttentionLegacy :356-372: per head h the qkv channels are
        // [q_h | k_h | v_h]; q and k are each scaled by ch^-1/4
        const int B = x.B, HW = x.H * x.W, C = a.ch, dh = C / a.heads;
        const long long M = (long long)B * HW;
        T4 out = alloc_t(ctx, B, x.H, x.W, C);
        const size_t mk = ctx.ws.mark();
        float* xn = ctx.ws.alloc_f((size_t)M * C);
        const bool sp = split_for_gemm(ctx, C);
        launch_groupnorm(ctx, x.p, C, C, nullptr, 0, 0, B, HW, 32, a.ng, a.nb, 1e-5f, 0, xn, sp);
        float* qkv = ctx.ws.alloc_f((size_t)M * 3 * C);
        linear_into(ctx, xn, C, M, C, a.qkv, nullptr, 0, qkv, 3 * C, 0, 0, sp ? M : 0);
        float* o = ctx.ws.alloc_f((size_t)M * C);
        const float sc = 1.0f / std::sqrt(std::sqrt((float)dh));
        attention_into(ctx, qkv, 3 * C, 3 * dh, qkv + dh, 3 * C, 3 * dh, qkv + 2 * dh, 3 * C, 3 * dh, B, a.heads, dh, HW,
                       HW, sc * sc, o, C);
        linear_into(ctx, o, C, M, C, a.proj, x.p, C, out.p, C);
        ctx.ws.release(mk);
        return out;
    }

    // shared (in/out): h holds one half of a guided step's batch; cleared by the SpatialTransformer that parts the halves
    T4 run_layers(Ctx& ctx, const std::vector<Layer>& layers, T4 h, const T4* skip, const float* emb_out, bool* shared = nullptr) {
        bool first = true;
        for (const Layer& l : layers) {
            const T4* x2 = first ? skip : nullptr;
            switch (l.kind) {
                case kConv: {
                    T4 o = alloc_t(ctx, h.B, h.H, h.W, convs[l.idx].cout);
                    ConvOpt co;
                    co.KH = co.KW = 3;
                    co.pad = 1;
                    conv_into(ctx, h, nullptr, convs[l.idx].w, co, o);
                    h = o;
                    break;
                }
                case kRes:
                    h = run_res(ctx, res[l.idx], h, x2, emb_out);
                    break;
                case kST:
                    h = run_st(ctx, st[l.idx], h, shared && *shared, exporting ? 2 : 0);
                    exporting = false;
                    if (shared) *shared = false;
                    break;
                case kAttn:
                    h = run_attn(ctx, attn[l.idx], h);
                    break;
                case kDown: {   // Downsample: conv3x3 stride 2 pad 1 (openaimodel.py:151-153)
                    T4 o = alloc_t(ctx, h.B, (h.H + 2 - 3) / 2 + 1, (h.W + 2 - 3) / 2 + 1, convs[l.idx].cout);
                    ConvOpt co;
                    co.KH = co.KW = 3;
                    co.pad = 1;
                    co.stride = 2;
                    conv_into(ctx, h, nullptr, convs[l.idx].w, co, o);
                    h = o;
                    break;
                }
                case kUp: {     // Upsample: nearest 2x then conv3x3 (:116-118), the gather reads through the upsample
                    T4 o = alloc_t(ctx, h.B, h.H * 2, h.W * 2, convs[l.idx].cout);
                    if (!conv_up2_into(ctx, h, convs[l.idx].w_up2, o)) {      // (exact-fp32 mode, odd channel counts, MAA_PP=off)
                        ConvOpt co;
                        co.KH = co.KW = 3;
                        co.pad = 1;
                        co.up = 1;
                        conv_into(ctx, h, nullptr, convs[l.idx].w, co, o);
                    }
                    h = o;
                    break;
                }
            }
            first = false;
        }
        return h;
    }

    // rows = one per timestep: emb_layers(SiLU(time_embed(timestep_embedding(t)))) for every ResBlock (openaimodel.py:725-726,
    // 218-224, 264) -> out [rows, emb_all.Npad].  `add` (I2A: context.squeeze(1), custom_openaimodel.py:352-354) is a residual
    // of the second time_embed Linear.
    void emb_rows(Ctx& ctx, const float* t, int rows, const float* add, float* out) {
        const int mc = cfg.model_channels;
        float* te = ctx.ws.alloc_f((size_t)rows * mc);
        launch_timestep_embedding(ctx, t, rows, mc, te);
        float* e1 = ctx.ws.alloc_f((size_t)rows * emb_dim);
        linear_into(ctx, te, mc, rows, mc, te0, nullptr, 0, e1, emb_dim);
        float* emb = ctx.ws.alloc_f((size_t)rows * emb_dim);
        launch_silu(ctx, e1, (long long)rows * emb_dim, e1);
        linear_into(ctx, e1, emb_dim, rows, emb_dim, te2, add, emb_dim, emb, emb_dim);
        float* semb = ctx.ws.alloc_f((size_t)rows * emb_dim);
        launch_silu(ctx, emb, (long long)rows * emb_dim, semb);
        linear_into(ctx, semb, emb_dim, rows, emb_dim, emb_all, nullptr, 0, out, emb_all.Npad);
    }
    bool exporting = false;      // the next SpatialTransformer run_layers meets hands its prefix over (share 2)
    int emb_ld = 0;      // pitch between the samples' rows of the current forward's emb_out (0: one row for every sample)
    // a forward over samples [batch_off, batch_off + B) of the batch set_context saw (a lane of a CFG step: ddim.cpp); the
    // caller's x / t / out pointers already stand at that sample, the context rows and the K/V caches are offset here
    int batch_off = 0;
    bool lane = false;

    // emb_row != null: the step's ResBlock time-embedding row computed beforehand (UNet::emb_table), shared by all samples
    // share_req (guided steps, one time-embedding row for all samples = emb_row): 1 = the batch is cat([x] * 2) on one stream,
    // 2 / 3 = this call is the unconditional / conditional lane of such a step (the caller runs 2 before 3)
    void forward(Ctx& ctx, const float* x_nchw, const float* t, const float* context, int B, int H, int W,
                 float* out_nchw, const float* emb_row = nullptr, int share_req = 0) {
        if (context && batch_off) context += (size_t)batch_off * kv_len * cfg.context_dim;
        share = 0;
        exporting = false;      // (a forward that threw half-way must not leave its hand-over armed)
        if (emb_row) {
            emb_ld = 0;
            if (share_req == 1 && cfg.use_spatial_transformer && !lane && B % 2 == 0) share = 1;
            if ((share_req == 2 || share_req == 3) && lane && prefix_ok()) share = share_req;
            forward_body(ctx, x_nchw, B, H, W, out_nchw, emb_row);
            share = 0;
            return;
        }
        float* emb_out = ctx.ws.alloc_f((size_t)B * emb_all.Npad);
        emb_rows(ctx, t, B, cfg.add_context_to_emb ? context : nullptr, emb_out);
        emb_ld = emb_all.Npad;
        forward_body(ctx, x_nchw, B, H, W, out_nchw, emb_out);
    }

    void forward_body(Ctx& ctx, const float* x_nchw, int B, int H, int W, float* out_nchw, const float* emb_out) {
        if (share == 3) {
            forward_rest_of_lane(ctx, B, H, W, out_nchw, emb_out);
            return;
        }
        if (share == 2) {
            if (!prefix.ready) MAA_HIP(hipEventCreateWithFlags(&prefix.ready, hipEventDisableTiming));
            prefix.B = B;
            prefix.H = H;
            prefix.W = W;
            exporting = true;
        }
        // share 1: samples [B/2, B) repeat samples [0, B/2) (x and the embedding row); only the context rows differ, and the context
        // enters at the first cross-attention -- until then one half is computed and the skip tensors are written twice
        bool shared = share == 1;
        auto both = [&](const T4& t) {
            T4 d = alloc_t(ctx, 2 * t.B, t.H, t.W, t.C);
            launch_dup_half(ctx, t.p, (long long)t.B * t.H * t.W * t.C, d.p);
            return d;
        };
        T4 h = alloc_t(ctx, shared ? B / 2 : B, H, W, cfg.in_channels);
        launch_nchw_to_nhwc(ctx, x_nchw, h.B, cfg.in_channels, H * W, h.p);
        std::vector<T4> hs;
        for (auto& blk : input) {
            h = run_layers(ctx, blk, h, nullptr, emb_out, &shared);
            hs.push_back(shared ? both(h) : h);
            if (share == 2 && hs.size() == 1) prefix.hs0 = h.p;      // conv_in's output: the other lane's first skip tensor
        }
        exporting = false;
        if (shared) {          // (no transformer on the way down: the halves part at the middle block at the latest)
            h = both(h);
            shared = false;
        }
        finish(ctx, h, hs, B, H, W, out_nchw, emb_out);
    }

    // The conditional lane of a guided step (share 3): conv_in, the first ResBlock and the first transformer up to attn2's
    // queries are the unconditional lane's -- same x, same embedding row -- and are read from its workspace.
    void forward_rest_of_lane(Ctx& ctx, int B, int H, int W, float* out_nchw, const float* emb_out) {
        MAA_CHECK(prefix.B == B && prefix.H == H && prefix.W == W && prefix.ready, "guided step: the lanes disagree on the shared prefix");
        if (!ctx.ws.dry) MAA_HIP(hipStreamWaitEvent(ctx.stream, prefix.ready, 0));
        const STW& s0 = st[input[1][1].idx];
        auto view = [&](const float* p, int C) {
            T4 t;
            t.B = B;
            t.H = H;
            t.W = W;
            t.C = C;
            t.p = const_cast<float*>(p);
            return t;
        };
        std::vector<T4> hs;
        hs.push_back(view(prefix.hs0, cfg.model_channels));
        T4 h = run_st(ctx, s0, view(prefix.xres, s0.ch), false, 3);
        hs.push_back(h);
        for (size_t i = 2; i < input.size(); ++i) {
            h = run_layers(ctx, input[i], h, nullptr, emb_out);
            hs.push_back(h);
        }
        finish(ctx, h, hs, B, H, W, out_nchw, emb_out);
    }

    // middle block, the way up with the skip tensors, the output head
    void finish(Ctx& ctx, T4 h, std::vector<T4>& hs, int B, int H, int W, float* out_nchw, const float* emb_out) {
        const int mc = cfg.model_channels;
        h = run_layers(ctx, middle, h, nullptr, emb_out);
        for (auto& blk : output) {
            T4 skip = hs.back();
            hs.pop_back();
            h = run_layers(ctx, blk, h, &skip, emb_out);
        }
        T4 hn = alloc_t(ctx, B, H, W, mc);
        hn.split = split_for_gemm(ctx, mc);
        launch_groupnorm(ctx, h.p, mc, mc, nullptr, 0, 0, B, H * W, 32, out_g, out_b, 1e-5f, 1, hn.p, hn.split);
        if (hn.split && ctx.tune.up2 &&
            launch_narrow_conv3x3(ctx, hn.p, mc, B, H, W, mc, out_narrow.w, out_narrow.bias, cfg.out_channels, out_nchw))
            return;      // (bf16x3: the 320 -> 4 convolution on its own kernel, straight to NCHW)
        T4 o = alloc_t(ctx, B, H, W, cfg.out_channels);
        ConvOpt co;
        co.KH = co.KW = 3;
        co.pad = 1;
        conv_into(ctx, hn, nullptr, out_conv, co, o);
        launch_nhwc_to_nchw(ctx, o.p, B, cfg.out_channels, H * W, out_nchw, cfg.out_channels);
    }
};

UNet::UNet(const maa_unet_config& cfg, const StateDict& sd, int precision) : impl_(new Impl(precision)) {
    impl_->cfg = cfg;
    impl_->build(sd);
}
UNet::~UNet() { delete impl_; }
const maa_unet_config& UNet::config() const { return impl_->cfg; }
size_t UNet::weight_bytes() const { return impl_->ws.bytes(); }

void UNet::set_context(Ctx& ctx, const float* context, int B, int L) {
    Impl& m = *impl_;
    if (!m.cfg.use_spatial_transformer) return;
    PrecisionGuard pg(ctx, m.precision);
    const size_t rows = (size_t)B * L;
    if (rows > m.kv_cap_rows) {
        MAA_HIP(hipStreamSynchronize(ctx.stream));      // the old caches may still be read by queued launches
        for (size_t i = 0; i < m.kv_cache.size(); ++i) {
            if (m.kv_cache[i]) MAA_HIP(hipFree(m.kv_cache[i]));
            m.kv_cache[i] = nullptr;
            void* d = nullptr;
            MAA_HIP(hipMalloc(&d, rows * 2 * m.kv_inner[i] * sizeof(float)));
            m.kv_cache[i] = static_cast<float*>(d);
        }
        m.kv_cap_rows = rows;
    }
    for (const STW& s : m.st)
        for (const STBlockW& b : s.blocks)
            linear_into(ctx, context, m.cfg.context_dim, (long long)rows, m.cfg.context_dim, b.kv2, nullptr, 0,
                        m.kv_cache[b.kv_slot], 2 * s.heads * s.dh);
    m.kv_batch = B;
    m.kv_len = L;
    context_ptr = context;
}

void UNet::graph_key(std::vector<unsigned long long>& key) const {
    const Impl& m = *impl_;
    key.push_back((unsigned long long)reinterpret_cast<uintptr_t>(this));
    key.push_back(m.serial);                 // (a later UNet may be constructed at a freed one's address)
    key.push_back((unsigned long long)reinterpret_cast<uintptr_t>(context_ptr));
    key.push_back((unsigned long long)m.kv_batch);
    key.push_back((unsigned long long)m.kv_len);
    for (const float* p : m.kv_cache) key.push_back((unsigned long long)reinterpret_cast<uintptr_t>(p));
}

void UNet::set_context_cfg(Ctx& ctx, const float* d_uncond, const float* d_cond, int B, int L) {
    Impl& m = *impl_;
    const size_t half = (size_t)B * L * m.cfg.context_dim * sizeof(float);
    char* buf = static_cast<char*>(m.cfg_context.get(2 * half, ctx.stream));
    MAA_HIP(hipMemcpyAsync(buf, d_uncond, half, hipMemcpyDeviceToDevice, ctx.stream));
    MAA_HIP(hipMemcpyAsync(buf + half, d_cond, half, hipMemcpyDeviceToDevice, ctx.stream));
    set_context(ctx, reinterpret_cast<const float*>(buf), 2 * B, L);
}

void UNet::forward(Ctx& ctx, const float* x_nchw, const float* t, const float* context, int B, int H, int W,
                   float* out_nchw, const float* emb_row, int batch_off, int cfg_share) {
    Impl& m = *impl_;
    PrecisionGuard pg(ctx, m.precision);
    m.lane = batch_off >= 0;
    m.batch_off = m.lane ? batch_off : 0;
    try {
        run_sized(ctx, [&] { m.forward(ctx, x_nchw, t, context, B, H, W, out_nchw, emb_row, cfg_share); });
    } catch (...) {
        m.batch_off = 0;
        m.lane = false;
        throw;
    }
    m.batch_off = 0;
    m.lane = false;
}
int UNet::emb_width() const { return impl_->emb_all.Npad; }
void UNet::emb_table(Ctx& ctx, const float* d_t, int rows, float* d_out) {
    Impl& m = *impl_;
    MAA_CHECK(!m.cfg.add_context_to_emb, "emb_table: this UNet's embedding depends on the sample's context");
    PrecisionGuard pg(ctx, m.precision);
    run_sized(ctx, [&] { m.emb_rows(ctx, d_t, rows, nullptr, d_out); });
}

}  // namespace maa
