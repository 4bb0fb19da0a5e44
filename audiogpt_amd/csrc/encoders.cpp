// Conditioning encoders on the device: the two towers whose outputs are the sampler's cross-attention
// context, so that a prompt goes text / image -> context -> latents -> mel -> waveform without leaving the GPU.
//
//   kind 0  CLAP text branch as FrozenCLAPEmbedder.encode runs it (ldm/modules/encoders/modules.py:204-211):
//           BertModel(input_ids) -- transformers' bert-base-uncased, post-LayerNorm encoder, erf GELU, no attention mask
//           (the reference passes input_ids only) -- then Projection on every token (CLAP/clap.py:8-20):
//           e1 = linear1(h); LayerNorm(e1 + linear2(gelu(e1)))
//   kind 1  OpenCLIP image tower behind FrozenGlobalNormOpenCLIPEmbedder.forward_img (modules.py:340-343): open_clip's
//           VisionTransformer (conv1 patchify, class token, positional embedding, ln_pre, pre-LayerNorm residual blocks
//           with nn.MultiheadAttention and a c_fc / GELU / c_proj MLP, ln_post on the class token, @ proj), then
//           z / ||z||.  open_clip itself is not part of the reference tree (a pip dependency): its published
//           architecture is what is restated here and in oracle/encoders.py.
//   kind 2  OpenCLIP text tower behind FrozenGlobalNormOpenCLIPEmbedder.forward (modules.py:334-338; the image-to-audio
//           tool encodes its unconditional prompt "" with it, audio-chatgpt.py:238): token + positional embedding, the
//           same pre-LayerNorm blocks under a causal mask, ln_final, the features at the end-of-text token (argmax of
//           the ids) @ text_projection, then z / ||z||.
//
// Both are sequences of the engines the UNet already uses: a Linear is an igemm with bias / GELU / residual epilogues,
// the head-major attention is flash_attn.hip (d = 64 / 80) in the bf16 modes and the three-launch path in exact fp32,
// LayerNorm writes split32 rows where its only consumer is a contraction.
#include "models.h"

#include <cmath>

namespace maa {

void launch_bert_embed(const Ctx& ctx, const int* ids, long long rows, int L, int C, int vocab, const float* word,
                       const float* pos, const float* type0, float* out);
void launch_vit_tokens(const Ctx& ctx, const float* patches, int B, int P, int C, const float* cls, const float* pos,
                       float* out);
void launch_gather_rows(const Ctx& ctx, const float* x, long long stride, int B, int C, float* out);
void launch_l2norm_rows(const Ctx& ctx, const float* x, int B, int C, float* out);
void launch_gather_argmax_rows(const Ctx& ctx, const int* ids, int B, int L, const float* x, int C, float* out);
void launch_gelu(const Ctx& ctx, const float* x, long long n, float* out);

struct Encoder::Impl {
    maa_encoder_config cfg;
    int precision = 0;
    WeightStore ws;
    explicit Impl(int prec) : precision(prec), ws(prec != 0) {}

    struct Layer {
        PackedW qkv, out, fc1, fc2;
        float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    };
    std::vector<Layer> layers;
    // text
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *emb_g = nullptr, *emb_b = nullptr;
    PackedW proj1, proj2;
    float *proj_g = nullptr, *proj_b = nullptr;
    // image
    PackedW patch, vproj;
    float *cls = nullptr, *vpos = nullptr, *pre_g = nullptr, *pre_b = nullptr, *post_g = nullptr, *post_b = nullptr;

    void build_text(const StateDict& sd) {
        const std::string e = "base.embeddings.";
        MAA_CHECK(get(sd, e + "word_embeddings.weight").shape[0] == cfg.vocab, "BERT vocabulary size");
        MAA_CHECK(get(sd, e + "position_embeddings.weight").shape[0] >= cfg.max_positions, "BERT position table");
        word = ws.vec(sd, e + "word_embeddings.weight");
        pos = ws.vec(sd, e + "position_embeddings.weight");
        type0 = ws.vec(sd, e + "token_type_embeddings.weight");      // row 0 is the one used (token_type_ids = 0)
        emb_g = ws.vec(sd, e + "LayerNorm.weight");
        emb_b = ws.vec(sd, e + "LayerNorm.bias");
        for (int i = 0; i < cfg.layers; ++i) {
            const std::string p = "base.encoder.layer." + std::to_string(i) + ".";
            Layer l;
            l.qkv = ws.pack_concat(sd, {p + "attention.self.query.weight", p + "attention.self.key.weight", p + "attention.self.value.weight"},
                                   {p + "attention.self.query.bias", p + "attention.self.key.bias", p + "attention.self.value.bias"});
            l.out = ws.pack_conv(sd, p + "attention.output.dense.weight", p + "attention.output.dense.bias", 1, 1);
            l.ln1g = ws.vec(sd, p + "attention.output.LayerNorm.weight");
            l.ln1b = ws.vec(sd, p + "attention.output.LayerNorm.bias");
            l.fc1 = ws.pack_conv(sd, p + "intermediate.dense.weight", p + "intermediate.dense.bias", 1, 1);
            l.fc2 = ws.pack_conv(sd, p + "output.dense.weight", p + "output.dense.bias", 1, 1);
            l.ln2g = ws.vec(sd, p + "output.LayerNorm.weight");
            l.ln2b = ws.vec(sd, p + "output.LayerNorm.bias");
            MAA_CHECK(l.qkv.N == 3 * cfg.width && l.fc1.N == cfg.mlp_dim, "BERT layer widths");
            layers.push_back(l);
        }
        proj1 = ws.pack_conv(sd, "projection.linear1.weight", "", 1, 1);
        proj2 = ws.pack_conv(sd, "projection.linear2.weight", "", 1, 1);
        proj_g = ws.vec(sd, "projection.layer_norm.weight");
        proj_b = ws.vec(sd, "projection.layer_norm.bias");
        MAA_CHECK(proj1.N == cfg.d_proj && proj2.N == cfg.d_proj && proj1.K == cfg.width, "CLAP projection widths");
    }

    // x @ proj with proj [width, d_proj]: as a Linear its weight is proj^T
    PackedW pack_right_matrix(const StateDict& sd, const std::string& name) {
        const HostTensor& pr = get(sd, name);
        MAA_CHECK(pr.shape.size() == 2 && pr.shape[0] == cfg.width && pr.shape[1] == cfg.d_proj, name + " shape");
        std::vector<float> t((size_t)cfg.width * cfg.d_proj);
        for (int k = 0; k < cfg.width; ++k)
            for (int n = 0; n < cfg.d_proj; ++n) t[(size_t)n * cfg.width + k] = pr.data[(size_t)k * cfg.d_proj + n];
        StateDict tmp;
        HostTensor h;
        h.data = t.data();
        h.shape = {cfg.d_proj, cfg.width};
        tmp["w"] = h;
        return ws.pack_conv(tmp, "w", "", 1, 1);
    }

    void build_resblocks(const StateDict& sd) {
        for (int i = 0; i < cfg.layers; ++i) {
            const std::string p = "transformer.resblocks." + std::to_string(i) + ".";
            Layer l;
            l.ln1g = ws.vec(sd, p + "ln_1.weight");
            l.ln1b = ws.vec(sd, p + "ln_1.bias");
            l.qkv = ws.pack_conv(sd, p + "attn.in_proj_weight", p + "attn.in_proj_bias", 1, 1);     // rows [q | k | v]
            l.out = ws.pack_conv(sd, p + "attn.out_proj.weight", p + "attn.out_proj.bias", 1, 1);
            l.ln2g = ws.vec(sd, p + "ln_2.weight");
            l.ln2b = ws.vec(sd, p + "ln_2.bias");
            l.fc1 = ws.pack_conv(sd, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", 1, 1);
            l.fc2 = ws.pack_conv(sd, p + "mlp.c_proj.weight", p + "mlp.c_proj.bias", 1, 1);
            MAA_CHECK(l.qkv.N == 3 * cfg.width && l.fc1.N == cfg.mlp_dim, "OpenCLIP layer widths");
            layers.push_back(l);
        }
    }

    void build_clip_text(const StateDict& sd) {
        MAA_CHECK(get(sd, "token_embedding.weight").shape[0] == cfg.vocab, "OpenCLIP vocabulary size");
        MAA_CHECK(get(sd, "positional_embedding").numel() == (long long)cfg.max_positions * cfg.width, "OpenCLIP text positions");
        word = ws.vec(sd, "token_embedding.weight");
        pos = ws.vec(sd, "positional_embedding");
        build_resblocks(sd);
        post_g = ws.vec(sd, "ln_final.weight");
        post_b = ws.vec(sd, "ln_final.bias");
        vproj = pack_right_matrix(sd, "text_projection");
    }

    void build_image(const StateDict& sd) {
        patch = ws.pack_conv(sd, "conv1.weight", "", cfg.patch, cfg.patch);
        MAA_CHECK(patch.N == cfg.width && patch.K == 3 * cfg.patch * cfg.patch, "ViT patch embedding shape");
        cls = ws.vec(sd, "class_embedding");
        const int tokens = (cfg.image / cfg.patch) * (cfg.image / cfg.patch) + 1;
        MAA_CHECK(get(sd, "positional_embedding").numel() == (long long)tokens * cfg.width, "ViT positional embedding shape");
        vpos = ws.vec(sd, "positional_embedding");
        pre_g = ws.vec(sd, "ln_pre.weight");
        pre_b = ws.vec(sd, "ln_pre.bias");
        build_resblocks(sd);
        post_g = ws.vec(sd, "ln_post.weight");
        post_b = ws.vec(sd, "ln_post.bias");
        vproj = pack_right_matrix(sd, "proj");
    }

    // self-attention over [B, L, W] rows given as fused qkv [rows, 3W] (columns [q | k | v], heads contiguous inside each)
    void attend(Ctx& ctx, const float* qkv, int B, int L, float* o, int o_split, int causal = 0) {
        const int W = cfg.width, dh = W / cfg.heads;
        attention_into(ctx, qkv, 3 * W, dh, qkv + W, 3 * W, dh, qkv + 2 * W, 3 * W, dh, B, cfg.heads, dh, L, L,
                       1.0f / std::sqrt((float)dh), o, W, o_split, causal);
    }

    // pre-LayerNorm residual blocks shared by the two OpenCLIP towers: x = x + attn(ln_1(x)); x = x + mlp(ln_2(x)).
    // x, y: [M, W] fp32 (the result is left in x)
    void resblocks(Ctx& ctx, float* x, float* y, int B, int L, int causal) {
        const int W = cfg.width, F = cfg.mlp_dim;
        const long long M = (long long)B * L;
        float* ln = ctx.ws.alloc_f((size_t)M * W);
        float* qkv = ctx.ws.alloc_f((size_t)M * 3 * W);
        float* o = ctx.ws.alloc_f((size_t)M * W);
        float* f = ctx.ws.alloc_f((size_t)M * F);
        const int sp = split_for_gemm(ctx, W) ? 1 : 0;
        const int o_sp = sp && flash_attention_covers(ctx, W / cfg.heads) ? 1 : 0;
        const int f_sp = split_for_gemm(ctx, F) ? 1 : 0;
        for (const Layer& l : layers) {
            launch_layernorm(ctx, x, M, W, l.ln1g, l.ln1b, cfg.ln_eps, ln, sp);
            linear_into(ctx, ln, W, M, W, l.qkv, nullptr, 0, qkv, 3 * W, 0, 0, sp ? M : 0);
            attend(ctx, qkv, B, L, o, o_sp, causal);
            linear_into(ctx, o, W, M, W, l.out, x, W, y, W, 0, 0, o_sp ? M : 0);
            launch_layernorm(ctx, y, M, W, l.ln2g, l.ln2b, cfg.ln_eps, ln, sp);
            linear_into(ctx, ln, W, M, W, l.fc1, nullptr, 0, f, F, 0, 0, sp ? M : 0, f_sp, /*act=*/3);
            linear_into(ctx, f, F, M, F, l.fc2, y, W, x, W, 0, 0, f_sp ? M : 0);
        }
    }

    // open_clip CLIP.encode_text + L2 normalisation: ids [B, L] -> [B, d_proj]
    void clip_text(Ctx& ctx, const int* ids, int B, int L, float* out) {
        const int W = cfg.width, D = cfg.d_proj;
        const long long M = (long long)B * L;
        float* x = ctx.ws.alloc_f((size_t)M * W);
        float* y = ctx.ws.alloc_f((size_t)M * W);
        launch_bert_embed(ctx, ids, M, L, W, cfg.vocab, word, pos, nullptr, x);
        resblocks(ctx, x, y, B, L, /*causal=*/1);
        float* c = ctx.ws.alloc_f((size_t)B * W);
        float* cn = ctx.ws.alloc_f((size_t)B * W);
        float* z = ctx.ws.alloc_f((size_t)B * D);
        launch_gather_argmax_rows(ctx, ids, B, L, x, W, c);             // LayerNorm is per row: gather first, then ln_final
        launch_layernorm(ctx, c, B, W, post_g, post_b, cfg.ln_eps, cn);
        linear_into(ctx, cn, W, B, W, vproj, nullptr, 0, z, D);
        launch_l2norm_rows(ctx, z, B, D, out);
    }

    // BertLayer x 12 (post-LN): h = LN(h + dense(attn(h))); h = LN(h + dense(gelu(dense(h))))
    // cls = true: the scorer's TextEncoder (wav_evaluation/models/clap.py:49-53): Projection of the [CLS] row only, then
    // CLAPWrapper's unit-length normalisation -> [B, d_proj].  (The scorer feeds BERT a padded sequence with its
    // attention mask; masked keys get an additive finfo.min, i.e. exactly zero weight, so running the unpadded ids --
    // what the caller passes here -- gives the same [CLS] row.)
    void text(Ctx& ctx, const int* ids, int B, int L, float* out, bool cls = false) {
        const int W = cfg.width, F = cfg.mlp_dim, D = cfg.d_proj;
        const long long M = (long long)B * L;
        float* x = ctx.ws.alloc_f((size_t)M * W);
        float* h = ctx.ws.alloc_f((size_t)M * W);
        launch_bert_embed(ctx, ids, M, L, W, cfg.vocab, word, pos, type0, x);
        launch_layernorm(ctx, x, M, W, emb_g, emb_b, cfg.ln_eps, h);
        float* qkv = ctx.ws.alloc_f((size_t)M * 3 * W);
        float* o = ctx.ws.alloc_f((size_t)M * W);
        float* t = ctx.ws.alloc_f((size_t)M * W);
        float* f = ctx.ws.alloc_f((size_t)M * F);
        const int o_sp = split_for_gemm(ctx, W) && flash_attention_covers(ctx, W / cfg.heads) ? 1 : 0;
        const int f_sp = split_for_gemm(ctx, F) ? 1 : 0;
        for (const Layer& l : layers) {
            linear_into(ctx, h, W, M, W, l.qkv, nullptr, 0, qkv, 3 * W);
            attend(ctx, qkv, B, L, o, o_sp);
            linear_into(ctx, o, W, M, W, l.out, h, W, t, W, 0, 0, o_sp ? M : 0);
            launch_layernorm(ctx, t, M, W, l.ln1g, l.ln1b, cfg.ln_eps, x);
            linear_into(ctx, x, W, M, W, l.fc1, nullptr, 0, f, F, 0, 0, 0, f_sp, /*act=*/3);
            linear_into(ctx, f, F, M, F, l.fc2, x, W, t, W, 0, 0, f_sp ? M : 0);
            launch_layernorm(ctx, t, M, W, l.ln2g, l.ln2b, cfg.ln_eps, h);
        }
        // Projection on every token (FrozenCLAPEmbedder) or on the [CLS] rows (the scorer)
        const long long R = cls ? B : M;
        const float* src = h;
        if (cls) {
            float* c = ctx.ws.alloc_f((size_t)B * W);
            launch_gather_rows(ctx, h, (long long)L * W, B, W, c);
            src = c;
        }
        float* e1 = ctx.ws.alloc_f((size_t)R * D);
        float* g = ctx.ws.alloc_f((size_t)R * D);
        float* e2 = ctx.ws.alloc_f((size_t)R * D);
        linear_into(ctx, src, W, R, W, proj1, nullptr, 0, e1, D);
        launch_gelu(ctx, e1, R * D, g);
        linear_into(ctx, g, D, R, D, proj2, e1, D, e2, D);
        if (cls) {
            float* ln = ctx.ws.alloc_f((size_t)R * D);
            launch_layernorm(ctx, e2, R, D, proj_g, proj_b, 1e-5f, ln);
            launch_l2norm_rows(ctx, ln, B, D, out);
        } else {
            launch_layernorm(ctx, e2, R, D, proj_g, proj_b, 1e-5f, out);
        }
    }

    // open_clip VisionTransformer.forward + L2 normalisation
    void image(Ctx& ctx, const float* img, int B, float* out) {
        const int W = cfg.width, D = cfg.d_proj, S = cfg.image, G = S / cfg.patch, P = G * G, L = P + 1;
        const long long M = (long long)B * L;
        float* nhwc = ctx.ws.alloc_f((size_t)B * S * S * 3);
        launch_nchw_to_nhwc(ctx, img, B, 3, S * S, nhwc);
        T4 a, pe;
        a.B = B;
        a.H = a.W = S;
        a.C = 3;
        a.p = nhwc;
        pe = alloc_t(ctx, B, G, G, W);
        ConvOpt co;
        co.KH = co.KW = cfg.patch;
        co.stride = cfg.patch;
        conv_into(ctx, a, nullptr, patch, co, pe);
        float* x = ctx.ws.alloc_f((size_t)M * W);
        float* y = ctx.ws.alloc_f((size_t)M * W);
        launch_vit_tokens(ctx, pe.p, B, P, W, cls, vpos, y);
        launch_layernorm(ctx, y, M, W, pre_g, pre_b, cfg.ln_eps, x);
        resblocks(ctx, x, y, B, L, /*causal=*/0);
        float* c = ctx.ws.alloc_f((size_t)B * W);
        float* cn = ctx.ws.alloc_f((size_t)B * W);
        float* z = ctx.ws.alloc_f((size_t)B * D);
        launch_gather_rows(ctx, x, (long long)L * W, B, W, c);
        launch_layernorm(ctx, c, B, W, post_g, post_b, cfg.ln_eps, cn);
        linear_into(ctx, cn, W, B, W, vproj, nullptr, 0, z, D);
        launch_l2norm_rows(ctx, z, B, D, out);
    }
};

Encoder::Encoder(const maa_encoder_config& cfg, const StateDict& sd, int precision) : impl_(new Impl(precision)) {
    impl_->cfg = cfg;
    try {
        if (cfg.kind == 0)
            impl_->build_text(sd);
        else if (cfg.kind == 1)
            impl_->build_image(sd);
        else
            impl_->build_clip_text(sd);
    } catch (...) {
        delete impl_;
        throw;
    }
}
Encoder::~Encoder() { delete impl_; }
const maa_encoder_config& Encoder::config() const { return impl_->cfg; }

void Encoder::text(Ctx& ctx, const int* d_ids, int B, int L, float* d_out) {
    MAA_CHECK(impl_->cfg.kind == 0 || impl_->cfg.kind == 2, "encoder_text on an image tower");
    MAA_CHECK(L <= impl_->cfg.max_positions, "sequence longer than the position table");
    PrecisionGuard guard(ctx, impl_->precision);
    if (impl_->cfg.kind == 0)
        run_sized(ctx, [&] { impl_->text(ctx, d_ids, B, L, d_out); });
    else
        run_sized(ctx, [&] { impl_->clip_text(ctx, d_ids, B, L, d_out); });
}

void Encoder::text_cls(Ctx& ctx, const int* d_ids, int B, int L, float* d_out) {
    MAA_CHECK(impl_->cfg.kind == 0, "encoder_text_cls is the CLAP (BERT) tower's");
    MAA_CHECK(L <= impl_->cfg.max_positions, "sequence longer than the position table");
    PrecisionGuard guard(ctx, impl_->precision);
    run_sized(ctx, [&] { impl_->text(ctx, d_ids, B, L, d_out, /*cls=*/true); });
}

void Encoder::image(Ctx& ctx, const float* d_img, int B, float* d_out) {
    MAA_CHECK(impl_->cfg.kind == 1, "encoder_image on a text tower");
    PrecisionGuard guard(ctx, impl_->precision);
    run_sized(ctx, [&] { impl_->image(ctx, d_img, B, d_out); });
}

}  // namespace maa
