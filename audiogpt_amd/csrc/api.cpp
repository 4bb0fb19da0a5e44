// extern "C" surface of libaudiogpt_mi355x (include/maa.h): argument checking, handle ownership and the
// exception -> status translation.  No C++ type or exception crosses this file's boundary.
#include "models.h"

#include <cstring>
#include <memory>

namespace maa {
const char* last_error_cstr();
}

struct maa_ctx {
    maa::Ctx c;
    bool owns_stream = false;
};
struct maa_unet {
    std::unique_ptr<maa::UNet> m;
};
struct maa_vae {
    std::unique_ptr<maa::VAE> m;
};
struct maa_vocoder {
    std::unique_ptr<maa::Vocoder> m;
};
struct maa_diffnet {
    std::unique_ptr<maa::DiffNet> m;
};
struct maa_encoder {
    std::unique_ptr<maa::Encoder> m;
};
struct maa_clap_audio {
    std::unique_ptr<maa::ClapAudio> m;
};
struct maa_spectral {
    std::unique_ptr<maa::Spectral> m;
};
struct maa_resampler {
    std::unique_ptr<maa::Resampler> m;
};

namespace {

template <class F>
int guarded(F&& f) {
    try {
        f();
        return MAA_OK;
    } catch (const maa::Error& e) {
        maa::set_last_error(e.what());
        const bool hip = std::strstr(e.what(), "hip") != nullptr;
        return hip ? MAA_ERR_HIP : MAA_ERR_INVALID;
    } catch (const std::exception& e) {
        maa::set_last_error(std::string("internal: ") + e.what());
        return MAA_ERR_INTERNAL;
    } catch (...) {
        maa::set_last_error("internal: unknown exception");
        return MAA_ERR_INTERNAL;
    }
}

maa::StateDict to_state_dict(const maa_tensor* tensors, int n) {
    maa::StateDict sd;
    MAA_CHECK(tensors != nullptr || n == 0, "null tensor list");
    for (int i = 0; i < n; ++i) {
        const maa_tensor& t = tensors[i];
        MAA_CHECK(t.name && t.data && t.ndim >= 0 && t.ndim <= 6, "malformed maa_tensor");
        maa::HostTensor h;
        h.data = t.data;
        for (int d = 0; d < t.ndim; ++d) h.shape.push_back(t.shape[d]);
        sd[t.name] = h;
    }
    return sd;
}

void bind(maa_ctx* ctx) {
    MAA_CHECK(ctx != nullptr, "null context");
    MAA_HIP(hipSetDevice(ctx->c.device));
}

// single-tensor state dict helpers for the operator entry points
struct OneShot {
    maa::StateDict sd;
    void add(const std::string& name, const float* data, std::vector<long long> shape) {
        maa::HostTensor h;
        h.data = data;
        h.shape = std::move(shape);
        sd[name] = h;
    }
};

}  // namespace

extern "C" {

const char* maa_last_error(void) { return maa::last_error_cstr(); }
const char* maa_version(void) { return "libaudiogpt_mi355x 0.1.0 gfx950"; }

int maa_ctx_create(int device_id, void* hip_stream, maa_ctx** out) {
    return guarded([&] {
        MAA_CHECK(out != nullptr, "null out");
        int n = 0;
        MAA_HIP(hipGetDeviceCount(&n));
        MAA_CHECK(device_id >= 0 && device_id < n, "no such HIP device");
        MAA_HIP(hipSetDevice(device_id));
        auto* c = new maa_ctx;
        c->c.device = device_id;
        if (hip_stream) {
            c->c.stream = static_cast<hipStream_t>(hip_stream);
        } else {
            // own stream, created WITHOUT hipStreamNonBlocking: it orders against the legacy default stream
            // (where PyTorch-ROCm puts its copies), and unlike the default stream it can be graph-captured
            hipStream_t s = nullptr;
            hipError_t e = hipStreamCreate(&s);
            if (e != hipSuccess) {
                delete c;
                throw maa::Error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
            }
            c->c.stream = s;
            c->owns_stream = true;
        }
        void* z = nullptr;
        if (hipMalloc(&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess) {
            delete c;
            throw maa::Error("hipMalloc: zero page");
        }
        c->c.zeros = static_cast<float*>(z);
        try {
            c->c.tune.load();
        } catch (...) {      // a malformed override: nothing of the half-built context survives the error
            (void)hipFree(z);
            if (c->owns_stream) (void)hipStreamDestroy(c->c.stream);
            delete c;
            throw;
        }
        *out = c;
    });
}
int maa_ctx_destroy(maa_ctx* ctx) {
    return guarded([&] {
        if (!ctx) return;
        (void)hipSetDevice(ctx->c.device);
        (void)hipStreamSynchronize(ctx->c.stream);
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->c.stream);
        if (ctx->c.zeros) (void)hipFree(ctx->c.zeros);
        delete ctx->c.prof;
        delete ctx;
    });
}
int maa_ctx_reload_tuning(maa_ctx* ctx) {
    return guarded([&] {
        bind(ctx);
        ctx->c.tune.load();
        ctx->c.ddim_graph.clear();      // a kept step graph was captured under the old knobs
    });
}
int maa_ctx_synchronize(maa_ctx* ctx) {
    return guarded([&] {
        bind(ctx);
        MAA_HIP(hipStreamSynchronize(ctx->c.stream));
    });
}
int maa_ctx_set_stream(maa_ctx* ctx, void* hip_stream) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(hip_stream != nullptr, "set_stream needs a non-default stream");
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->c.stream);
        ctx->owns_stream = false;
        ctx->c.stream = static_cast<hipStream_t>(hip_stream);
    });
}
int maa_ctx_workspace_bytes(maa_ctx* ctx, size_t* out) {
    return guarded([&] {
        MAA_CHECK(ctx && out, "null argument");
        // the second CFG lane (ddim.cpp) owns a workspace of its own, about half a batch's UNet arena, which is kept once it exists
        *out = ctx->c.ws.capacity() + (ctx->c.side ? ctx->c.side->ws.capacity() : 0);
    });
}

int maa_ctx_set_concurrency(maa_ctx* ctx, int n) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(n >= -1 && n != 0, "concurrency: -1 (guess from the live contexts) or the number of contexts kept in flight (>= 1)");
        if (ctx->c.kept_full() != (n >= 3))
            ctx->c.ddim_graph.clear();      // the kept step graph was captured under the other arrangement's launches
        ctx->c.concurrency = n;
    });
}
int maa_ctx_set_cfg_split(maa_ctx* ctx, int mode) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(mode >= -1 && mode <= 1, "cfg_split: -1 (default policy), 0 (one stream) or 1 (two lanes)");
        ctx->c.cfg_split = mode;
    });
}

int maa_ctx_set_precision(maa_ctx* ctx, int mode) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(mode >= 0 && mode <= 2, "precision mode must be 0 (fp32), 1 (bf16x3) or 2 (bf16)");
        ctx->c.dtype = mode;
    });
}

int maa_prof_begin(maa_ctx* ctx, int detail) {
    return guarded([&] {
        bind(ctx);
        if (!ctx->c.prof) ctx->c.prof = new maa::Profiler;
        ctx->c.prof->detail = detail;
        ctx->c.prof->pending.clear();
        ctx->c.prof->next = 0;
        if (ctx->c.side) {      // (the second lane of a CFG step gets its own table when a step first forks under profiling)
            delete ctx->c.side->prof;
            ctx->c.side->prof = nullptr;
        }
    });
}
int maa_prof_end(maa_ctx* ctx, maa_prof_row* rows, int max_rows, int* n_rows) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(ctx->c.prof && rows && n_rows && max_rows > 0, "prof_end without prof_begin / null output");
        auto agg = ctx->c.prof->collect(ctx->c.stream);
        delete ctx->c.prof;
        ctx->c.prof = nullptr;
        if (ctx->c.side && ctx->c.side->prof) {      // the launches of the second CFG lane, merged by name
            for (auto& r : ctx->c.side->prof->collect(ctx->c.side->stream)) {
                bool found = false;
                for (auto& a : agg)
                    if (a.name == r.name) {
                        a.launches += r.launches;
                        a.ms += r.ms;
                        a.flops += r.flops;
                        a.bytes += r.bytes;
                        found = true;
                        break;
                    }
                if (!found) agg.push_back(r);
            }
            delete ctx->c.side->prof;
            ctx->c.side->prof = nullptr;
        }
        int n = 0;
        for (auto& r : agg) {
            if (n >= max_rows) break;
            std::memset(&rows[n], 0, sizeof(maa_prof_row));
            std::strncpy(rows[n].name, r.name.c_str(), sizeof(rows[n].name) - 1);
            rows[n].launches = r.launches;
            rows[n].ms = r.ms;
            rows[n].flops = r.flops;
            rows[n].bytes = r.bytes;
            ++n;
        }
        *n_rows = n;
    });
}

// ------------------------------------------------------------------------------------------ UNet
int maa_unet_create(maa_ctx* ctx, const maa_unet_config* cfg, const maa_tensor* tensors, int n_tensors,
                    maa_unet** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(cfg && out, "null argument");
        MAA_CHECK(cfg->n_channel_mult > 0 && cfg->n_channel_mult <= 8 && cfg->model_channels % 32 == 0,
                  "unsupported UNet config");
        auto sd = to_state_dict(tensors, n_tensors);
        auto* u = new maa_unet;
        u->m.reset(new maa::UNet(*cfg, sd, ctx->c.dtype));
        *out = u;
    });
}
int maa_unet_destroy(maa_unet* u) {
    return guarded([&] { delete u; });
}
int maa_unet_set_context(maa_ctx* ctx, maa_unet* u, const float* d_context, int B, int L) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(u && d_context && B > 0 && L > 0, "bad set_context arguments");
        u->m->set_context(ctx->c, d_context, B, L);
    });
}
int maa_unet_forward(maa_ctx* ctx, maa_unet* u, const float* d_x, const float* d_t, int B, int H, int W,
                     float* d_out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(u && d_x && d_t && d_out && B > 0 && H > 0 && W > 0, "bad forward arguments");
        u->m->forward(ctx->c, d_x, d_t, u->m->context_ptr, B, H, W, d_out);
    });
}

int maa_ddim_update(maa_ctx* ctx, const float* d_x, const float* d_eps_uncond, const float* d_eps_cond, float scale,
                    const float* d_coef, int64_t n, float* d_x_prev, float* d_pred_x0) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_x && d_eps_uncond && d_coef && d_x_prev && n > 0, "bad ddim_update arguments");
        maa::launch_ddim_update(ctx->c, d_x, d_eps_uncond, d_eps_cond, scale, d_coef, n, d_x_prev, d_pred_x0);
    });
}
int maa_ddim_sample(maa_ctx* ctx, maa_unet* u, const maa_ddim_args* args, float* d_x) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(u && args && d_x && args->h_timesteps && args->h_alphas && args->h_alphas_prev, "bad ddim_sample arguments");
        maa::ddim_sample(ctx->c, *u->m, *args, d_x);
    });
}

// ------------------------------------------------------------------------------------------ VAE
int maa_vae_create(maa_ctx* ctx, const maa_vae_config* cfg, const maa_tensor* tensors, int n_tensors, maa_vae** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(cfg && out && cfg->n_ch_mult > 0 && cfg->n_ch_mult <= 8, "bad VAE config");
        auto sd = to_state_dict(tensors, n_tensors);
        auto* v = new maa_vae;
        v->m.reset(new maa::VAE(*cfg, sd, ctx->c.dtype));
        *out = v;
    });
}
int maa_vae_destroy(maa_vae* v) {
    return guarded([&] { delete v; });
}
int maa_vae_decode(maa_ctx* ctx, maa_vae* v, const float* d_z, int B, int h, int w, float inv_scale, float* d_mel) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(v && d_z && d_mel && B > 0 && h > 0 && w > 0, "bad vae_decode arguments");
        v->m->decode(ctx->c, d_z, B, h, w, inv_scale, d_mel);
    });
}
int maa_vae_decode_spec(maa_ctx* ctx, maa_vae* v, const float* d_z, int B, int h, int w, float inv_scale, float* d_spec) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(v && d_z && d_spec && B > 0 && h > 0 && w > 0, "bad vae_decode_spec arguments");
        v->m->decode_spec(ctx->c, d_z, B, h, w, inv_scale, d_spec);
    });
}
int maa_vae_encode_moments(maa_ctx* ctx, maa_vae* v, const float* d_mel, int B, int H, int W, float* d_moments) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(v && d_mel && d_moments && B > 0 && H > 0 && W > 0, "bad vae_encode arguments");
        v->m->encode_moments(ctx->c, d_mel, B, H, W, d_moments);
    });
}

// ------------------------------------------------------------------------------------------ vocoder
int maa_vocoder_create(maa_ctx* ctx, const maa_vocoder_config* cfg, const maa_tensor* tensors, int n_tensors,
                       maa_vocoder** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(cfg && out && cfg->n_upsamples > 0 && cfg->n_upsamples <= 8 && cfg->n_kernels > 0 &&
                      cfg->n_kernels <= 8 && cfg->n_dilations > 0 && cfg->n_dilations <= 8,
                  "bad vocoder config");
        auto sd = to_state_dict(tensors, n_tensors);
        auto* v = new maa_vocoder;
        v->m.reset(new maa::Vocoder(*cfg, sd, ctx->c.dtype));
        *out = v;
    });
}
int maa_vocoder_destroy(maa_vocoder* v) {
    return guarded([&] { delete v; });
}
int maa_vocoder_forward(maa_ctx* ctx, maa_vocoder* v, const float* d_mel, int B, int T, float* d_wav) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(v && d_mel && d_wav && B > 0 && T > 0, "bad vocoder_forward arguments");
        v->m->forward(ctx->c, d_mel, B, T, d_wav);
    });
}
int maa_vocoder_forward_f0(maa_ctx* ctx, maa_vocoder* v, const float* d_mel, const float* d_f0, const float* d_rand_ini,
                           const float* d_noise, int B, int T, float* d_wav) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(v && d_mel && d_f0 && d_rand_ini && d_noise && d_wav && B > 0 && T > 0, "bad vocoder_forward_f0 arguments");
        v->m->forward_f0(ctx->c, d_mel, d_f0, d_rand_ini, d_noise, B, T, d_wav);
    });
}

// ------------------------------------------------------------------------------------------ operators
// ------------------------------------------------------------------------------------------ DiffSinger
int maa_diffnet_create(maa_ctx* ctx, const maa_diffnet_config* cfg, const maa_tensor* tensors, int n_tensors,
                       maa_diffnet** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(cfg && out && cfg->in_dims > 0 && cfg->hidden_size > 0 && cfg->residual_layers > 0 &&
                      cfg->residual_channels > 0 && cfg->residual_channels % 2 == 0 && cfg->dilation_cycle_length > 0,
                  "bad DiffNet config");
        auto sd = to_state_dict(tensors, n_tensors);
        auto* d = new maa_diffnet;
        d->m.reset(new maa::DiffNet(*cfg, sd, ctx->c.dtype));
        *out = d;
    });
}
int maa_diffnet_destroy(maa_diffnet* d) {
    return guarded([&] { delete d; });
}
int maa_diffnet_forward(maa_ctx* ctx, maa_diffnet* d, const float* d_spec, const float* d_t, const float* d_cond, int B,
                        int T, float* d_eps) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d && d_spec && d_t && d_cond && d_eps && B > 0 && T > 0, "bad diffnet_forward arguments");
        d->m->forward(ctx->c, d_spec, d_t, d_cond, B, T, d_eps);
    });
}
int maa_plms_sample(maa_ctx* ctx, maa_diffnet* d, const maa_plms_args* args, float* d_x) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d && args && d_x, "bad plms_sample arguments");
        d->m->plms_sample(ctx->c, *args, d_x);
    });
}

// ------------------------------------------------------------------------------------------ conditioning encoders
int maa_encoder_create(maa_ctx* ctx, const maa_encoder_config* cfg, const maa_tensor* tensors, int n_tensors,
                       maa_encoder** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(cfg && out && cfg->kind >= 0 && cfg->kind <= 2 && cfg->layers > 0 && cfg->width > 0 && cfg->heads > 0 &&
                      cfg->width % cfg->heads == 0 && cfg->width % 4 == 0 && cfg->mlp_dim > 0 && cfg->d_proj > 0 &&
                      cfg->d_proj % 4 == 0 && cfg->ln_eps > 0.f,
                  "bad encoder config");
        if (cfg->kind != 1)
            MAA_CHECK(cfg->vocab > 0 && cfg->max_positions > 0, "bad text encoder config");
        else
            MAA_CHECK(cfg->patch > 0 && cfg->image > 0 && cfg->image % cfg->patch == 0, "bad image encoder config");
        auto sd = to_state_dict(tensors, n_tensors);
        auto* e = new maa_encoder;
        e->m.reset(new maa::Encoder(*cfg, sd, ctx->c.dtype));
        *out = e;
    });
}
int maa_encoder_destroy(maa_encoder* e) {
    return guarded([&] { delete e; });
}
int maa_encoder_text(maa_ctx* ctx, maa_encoder* e, const int* d_ids, int B, int L, float* d_out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(e && d_ids && d_out && B > 0 && L > 0, "bad encoder_text arguments");
        e->m->text(ctx->c, d_ids, B, L, d_out);
    });
}
int maa_encoder_image(maa_ctx* ctx, maa_encoder* e, const float* d_img, int B, float* d_out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(e && d_img && d_out && B > 0, "bad encoder_image arguments");
        e->m->image(ctx->c, d_img, B, d_out);
    });
}

int maa_encoder_text_cls(maa_ctx* ctx, maa_encoder* e, const int* d_ids, int B, int L, float* d_out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(e && d_ids && d_out && B > 0 && L > 0, "bad encoder_text_cls arguments");
        e->m->text_cls(ctx->c, d_ids, B, L, d_out);
    });
}

int maa_clap_audio_create(maa_ctx* ctx, const maa_clap_audio_config* cfg, const maa_tensor* tensors, int n_tensors,
                          maa_clap_audio** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(cfg && out && cfg->mel_bins > 0 && cfg->n_blocks > 0 && cfg->n_blocks <= 8 && cfg->out_emb > 0 &&
                      cfg->d_proj > 0 && cfg->d_proj % 4 == 0 && cfg->bn_eps > 0.f,
                  "bad clap_audio config");
        for (int i = 0; i < cfg->n_blocks; ++i) MAA_CHECK(cfg->channels[i] > 0, "bad clap_audio channel list");
        auto sd = to_state_dict(tensors, n_tensors);
        auto* a = new maa_clap_audio;
        a->m.reset(new maa::ClapAudio(*cfg, sd, ctx->c.dtype));
        *out = a;
    });
}
int maa_clap_audio_destroy(maa_clap_audio* a) {
    return guarded([&] { delete a; });
}
int maa_clap_audio_embed(maa_ctx* ctx, maa_clap_audio* a, const float* d_logmel, int B, int T, float* d_embedding,
                         float* d_z) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(a && d_logmel && d_z && B > 0 && T > 0, "bad clap_audio_embed arguments");
        a->m->embed(ctx->c, d_logmel, B, T, d_embedding, d_z);
    });
}
int maa_clap_similarity(maa_ctx* ctx, const float* d_audio, const float* d_text, int Na, int Nt, int D, float scale,
                        float* d_out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_audio && d_text && d_out && Na > 0 && Nt > 0 && D > 0, "bad clap_similarity arguments");
        maa::launch_similarity(ctx->c, d_audio, d_text, Na, Nt, D, scale, d_out);
    });
}

int maa_spectral_create(maa_ctx* ctx, const maa_spectral_config* cfg, const float* h_basis, const float* h_melw,
                        maa_spectral** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(cfg && h_basis && h_melw && out, "bad spectral arguments");
        MAA_CHECK(cfg->n_fft > 0 && cfg->n_fft % 4 == 0 && cfg->hop > 0 && cfg->hop % 4 == 0 && cfg->n_freq == cfg->n_fft / 2 + 1 &&
                      cfg->n_mels > 0 && (cfg->pad_mode == 0 || cfg->pad_mode == 1) && (cfg->power == 1 || cfg->power == 2) &&
                      (cfg->log_kind == 0 || cfg->log_kind == 1) && cfg->amin > 0.f && (cfg->out_layout == 0 || cfg->out_layout == 1),
                  "bad spectral config (n_fft and hop must be multiples of 4)");
        auto* s = new maa_spectral;
        s->m.reset(new maa::Spectral(*cfg, h_basis, h_melw));
        *out = s;
    });
}
int maa_spectral_destroy(maa_spectral* s) {
    return guarded([&] { delete s; });
}
int maa_spectral_forward(maa_ctx* ctx, maa_spectral* s, const float* d_wav, int B, int n, float* d_out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(s && d_wav && d_out && B > 0 && n > 0, "bad spectral_forward arguments");
        MAA_CHECK(s->m->config().pad_mode == 0 || n > s->m->config().n_fft / 2, "signal too short for reflect padding");
        s->m->forward(ctx->c, d_wav, B, n, d_out);
    });
}

int maa_resampler_create(maa_ctx* ctx, int orig, int neu, int width, int klen, const float* h_kernels, maa_resampler** out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(out && h_kernels && orig > 0 && neu > 0 && width > 0 && klen > 0, "bad resampler arguments");
        auto* r = new maa_resampler;
        r->m.reset(new maa::Resampler(orig, neu, width, klen, h_kernels));
        *out = r;
    });
}
int maa_resampler_destroy(maa_resampler* r) {
    return guarded([&] { delete r; });
}
int maa_resampler_forward(maa_ctx* ctx, maa_resampler* r, const float* d_wav, int B, int n, float* d_out) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(r && d_wav && d_out && B > 0 && n > 0, "bad resampler_forward arguments");
        r->m->forward(ctx->c, d_wav, B, n, d_out);
    });
}

// MAA_OP_PRESPLIT=1 (tests): hand the activation to the contraction in the split32 form a normalisation would
// have written, so the op entry points exercise the LDS-DMA engines too.
static bool op_presplit(const maa::Ctx& c, int channels, bool other_prologue) {
    return c.tune.op_presplit && c.dtype != 0 && channels % 32 == 0 && !other_prologue;
}

int maa_op_linear(maa_ctx* ctx, const float* d_a, int M, int K, const float* h_w, const float* h_bias, int N,
                  int geglu, float* d_y) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_a && h_w && d_y && M > 0 && K > 0 && N > 0, "bad op_linear arguments");
        OneShot s;
        s.add("w", h_w, {N, K});
        if (h_bias) s.add("b", h_bias, {N});
        maa::WeightStore ws(ctx->c.dtype != 0);
        maa::Ctx& c = ctx->c;
        MAA_CHECK(!geglu || h_bias, "geglu needs a bias");
        maa::PackedW pw = geglu ? ws.pack_geglu(s.sd, "w", "b") : ws.pack_conv(s.sd, "w", h_bias ? "b" : "", 1, 1);
        const bool pre = op_presplit(c, K, false);
        maa::run_sized(c, [&] {
            const float* a = d_a;
            if (pre) {
                float* sp = c.ws.alloc_f((size_t)M * K);
                maa::launch_split32_pack(c, d_a, M, K, sp);
                a = sp;
            }
            maa::linear_into(c, a, K, M, K, pw, nullptr, 0, d_y, geglu ? N / 2 : N, geglu, 0, pre ? M : 0);
        });
        MAA_HIP(hipStreamSynchronize(c.stream));
    });
}

int maa_op_conv(maa_ctx* ctx, const float* d_x, int B, int Cin, int H, int W, const float* h_w, const float* h_bias,
                int Cout, int KH, int KW, int stride, int pad, int dil, int upsample2, float leaky_slope, float* d_y,
                int Ho, int Wo) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_x && h_w && d_y, "bad op_conv arguments");
        OneShot s;
        s.add("w", h_w, {Cout, Cin, KH, KW});
        if (h_bias) s.add("b", h_bias, {Cout});
        maa::WeightStore ws(ctx->c.dtype != 0);
        maa::Ctx& c = ctx->c;
        maa::PackedW pw = ws.pack_conv(s.sd, "w", h_bias ? "b" : "", KH, KW);
        maa::PackedW pw4;      // Upsample + conv3x3: the four-phase form the models take in the bf16 modes (blocks.cpp conv_up2_into)
        if (upsample2 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 && leaky_slope == 0.f)
            pw4 = ws.pack_conv_up2(s.sd, "w", h_bias ? "b" : "");
        maa::run_sized(c, [&] {
            maa::T4 x = maa::alloc_t(c, B, H, W, Cin);
            maa::launch_nchw_to_nhwc(c, d_x, B, Cin, H * W, x.p);
            if (!pw4.w && op_presplit(c, Cin, leaky_slope != 0.f)) {
                maa::T4 xs = maa::alloc_t(c, B, H, W, Cin);
                maa::launch_split32_pack(c, x.p, (long long)B * H * W, Cin, xs.p);
                xs.split = true;
                x = xs;
            }
            if (x.split && Cout <= 4 && h_bias && KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 && !upsample2 && c.tune.up2) {
                maa::PackedW pn = ws.pack_narrow3x3(s.sd, "w", "b");      // the UNet's output convolution (unet.cpp forward_body)
                if (maa::launch_narrow_conv3x3(c, x.p, Cin, B, H, W, Cin, pn.w, pn.bias, Cout, d_y)) return;
            }
            maa::T4 y = maa::alloc_t(c, B, Ho, Wo, Cout);
            maa::ConvOpt o;
            o.KH = KH;
            o.KW = KW;
            o.stride = stride;
            o.pad = pad;
            o.dil = dil;
            o.up = upsample2;
            if (leaky_slope != 0.f) {
                o.a_act = 1;
                o.a_slope = leaky_slope;
            }
            if (!(pw4.w && maa::conv_up2_into(c, x, pw4, y))) maa::conv_into(c, x, nullptr, pw, o, y);
            maa::launch_nhwc_to_nchw(c, y.p, B, Cout, Ho * Wo, d_y, Cout);
        });
        MAA_HIP(hipStreamSynchronize(c.stream));
    });
}

int maa_calib(maa_ctx* ctx, int kind, double* out_value) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(out_value && kind >= 0 && kind <= 3, "bad calib arguments");
        *out_value = maa::calib_run(ctx->c, kind);
    });
}

int maa_op_groupnorm(maa_ctx* ctx, const float* d_x, int B, int C, int HW, const float* h_gamma, const float* h_beta,
                     float eps, int silu, float* d_y) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_x && h_gamma && h_beta && d_y, "bad op_groupnorm arguments");
        maa::WeightStore ws(ctx->c.dtype != 0);
        maa::Ctx& c = ctx->c;
        float* g = ws.upload(std::vector<float>(h_gamma, h_gamma + C));
        float* b = ws.upload(std::vector<float>(h_beta, h_beta + C));
        maa::run_sized(c, [&] {
            maa::T4 x = maa::alloc_t(c, B, 1, HW, C), y = maa::alloc_t(c, B, 1, HW, C);
            maa::launch_nchw_to_nhwc(c, d_x, B, C, HW, x.p);
            maa::launch_groupnorm(c, x.p, C, C, nullptr, 0, 0, B, HW, 32, g, b, eps, silu, y.p);
            maa::launch_nhwc_to_nchw(c, y.p, B, C, HW, d_y, C);
        });
        MAA_HIP(hipStreamSynchronize(c.stream));
    });
}

int maa_op_layernorm(maa_ctx* ctx, const float* d_x, int rows, int C, const float* h_gamma, const float* h_beta,
                     float eps, float* d_y) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_x && h_gamma && h_beta && d_y, "bad op_layernorm arguments");
        maa::WeightStore ws(ctx->c.dtype != 0);
        maa::Ctx& c = ctx->c;
        float* g = ws.upload(std::vector<float>(h_gamma, h_gamma + C));
        float* b = ws.upload(std::vector<float>(h_beta, h_beta + C));
        maa::launch_layernorm(c, d_x, rows, C, g, b, eps, d_y);
        MAA_HIP(hipStreamSynchronize(c.stream));
    });
}

int maa_op_attention(maa_ctx* ctx, const float* d_q, const float* d_k, const float* d_v, int B, int heads, int dh,
                     int Nq, int Nk, float alpha, float* d_y) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_q && d_k && d_v && d_y, "bad op_attention arguments");
        maa::Ctx& c = ctx->c;
        const int C = heads * dh;
        maa::run_sized(c, [&] {
            maa::attention_into(c, d_q, C, dh, d_k, C, dh, d_v, C, dh, B, heads, dh, Nq, Nk, alpha, d_y, C);
        });
        MAA_HIP(hipStreamSynchronize(c.stream));
    });
}

int maa_op_conv_transpose1d(maa_ctx* ctx, const float* d_x, int B, int Cin, int L, const float* h_w,
                            const float* h_bias, int Cout, int k, int stride, float leaky_slope, float* d_y) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_x && h_w && h_bias && d_y, "bad op_conv_transpose1d arguments");
        OneShot s;
        s.add("w", h_w, {Cin, Cout, k});
        s.add("b", h_bias, {Cout});
        maa::WeightStore ws(ctx->c.dtype != 0);
        maa::Ctx& c = ctx->c;
        const int pad = (k - stride) / 2, U = k / stride;
        maa::run_sized(c, [&] {
            maa::T4 x = maa::alloc_t(c, B, 1, L, Cin), y = maa::alloc_t(c, B, 1, L * stride, Cout);
            maa::launch_nchw_to_nhwc(c, d_x, B, Cin, L, x.p);
            for (int carry = 0; carry < 2; ++carry) {
                bool any = false;
                for (int r = 0; r < stride; ++r) any = any || ((r + pad) / stride == carry);
                if (!any) continue;
                int r_start = 0, r_count = 0;
                maa::PackedW pw;
                if (!c.ws.dry) pw = ws.pack_convtr_phase(s.sd, "w", "b", stride, pad, carry, &r_start, &r_count);
                maa::IGemm p;
                p.a1 = x.p;
                p.lda1 = Cin;
                p.C1 = Cin;
                p.Win = L;
                p.Wout = L;
                p.KW = U;
                p.pw = U - 1 - carry;
                if (leaky_slope != 0.f) {
                    p.a_act = 1;
                    p.a_slope = leaky_slope;
                }
                p.b = pw.w;
                p.ldb = pw.ld;
                p.b_nk = pw.nk;
                p.b_split = pw.split;
                p.M = B * L;
                p.K = U * Cin;
                p.N = r_count * Cout;
                p.bias = pw.bias;
                p.c = y.p + (long long)r_start * Cout;
                p.ldc = stride * Cout;
                if (!c.ws.dry) maa::launch_igemm(c, p);
            }
            maa::launch_nhwc_to_nchw(c, y.p, B, Cout, L * stride, d_y, Cout);
        });
        MAA_HIP(hipStreamSynchronize(c.stream));
    });
}

int maa_op_bench_conv(maa_ctx* ctx, int B, int H, int W, int Cin, int Cout, int taps, int pre_split, int iters,
                      float* ms_per_launch) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(ms_per_launch && iters > 0 && (taps == 1 || taps == 9), "bad op_bench_conv arguments");
        maa::Ctx& c = ctx->c;
        const int k = taps == 9 ? 3 : 1;
        std::vector<float> hw((size_t)Cout * Cin * taps), hb(Cout, 0.1f);
        unsigned s = 12345u;
        for (auto& v : hw) {
            s = s * 1664525u + 1013904223u;
            v = ((int)(s >> 9) % 2001 - 1000) * 1e-4f;
        }
        OneShot sd;
        sd.add("w", hw.data(), {Cout, Cin, k, k});
        sd.add("b", hb.data(), {Cout});
        maa::WeightStore ws(c.dtype != 0);
        maa::PackedW pw = ws.pack_conv(sd.sd, "w", "b", k, k);
        const size_t n_in = (size_t)B * H * W * Cin, n_out = (size_t)B * H * W * Cout;
        std::vector<float> hx(n_in);
        for (auto& v : hx) {
            s = s * 1664525u + 1013904223u;
            v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f;
        }
        if (pre_split && c.dtype != 0) {
            // rewrite the rows as split32 lines ([32 bf16 hi | 32 bf16 lo] per 32 channels), the GroupNorm output form
            MAA_CHECK(Cin % 32 == 0, "op_bench_conv: pre-split input needs Cin % 32 == 0");
            auto f2bf = [](float f) {
                unsigned u;
                std::memcpy(&u, &f, 4);
                u += 0x7fffu + ((u >> 16) & 1u);
                return (unsigned short)(u >> 16);
            };
            std::vector<float> packed(n_in);
            unsigned short* d = reinterpret_cast<unsigned short*>(packed.data());
            for (size_t g = 0; g < n_in / 32; ++g)
                for (int j = 0; j < 32; ++j) {
                    const float v = hx[g * 32 + j];
                    const unsigned short h = f2bf(v);
                    const unsigned hu = (unsigned)h << 16;
                    float hf;
                    std::memcpy(&hf, &hu, 4);
                    d[g * 64 + j] = h;
                    d[g * 64 + 32 + j] = f2bf(v - hf);
                }
            hx.swap(packed);
        }
        float* dx = ws.upload(hx);
        float* dy = ws.upload(std::vector<float>(n_out, 0.f));
        maa::T4 x, y;
        x.B = y.B = B;
        x.H = y.H = H;
        x.W = y.W = W;
        x.C = Cin;
        y.C = Cout;
        x.p = dx;
        y.p = dy;
        if (pre_split && c.dtype != 0) {
            x.split = true;
        }
        maa::ConvOpt o;
        o.KH = o.KW = k;
        o.pad = k / 2;
        // (the split-K engine borrows its slabs from the arena: size it with the warm-up launches)
        maa::run_sized(c, [&] {
            for (int i = 0; i < 3; ++i) maa::conv_into(c, x, nullptr, pw, o, y);
        });
        hipEvent_t e0, e1;
        MAA_HIP(hipEventCreate(&e0));
        MAA_HIP(hipEventCreate(&e1));
        MAA_HIP(hipEventRecord(e0, c.stream));
        for (int i = 0; i < iters; ++i) maa::conv_into(c, x, nullptr, pw, o, y);
        MAA_HIP(hipEventRecord(e1, c.stream));
        MAA_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        MAA_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_launch = ms / iters;
    });
}

int maa_op_snake_aa(maa_ctx* ctx, const float* d_x, int B, int C, int L, const float* h_alpha, const float* h_beta,
                    int logscale, float* d_y) {
    return guarded([&] {
        bind(ctx);
        MAA_CHECK(d_x && h_alpha && h_beta && d_y, "bad op_snake_aa arguments");
        maa::WeightStore ws(ctx->c.dtype != 0);
        maa::Ctx& c = ctx->c;
        std::vector<float> ha(C), hib(C);
        for (int i = 0; i < C; ++i) {
            float a = h_alpha[i], b = h_beta[i];
            if (logscale) {
                a = std::exp(a);
                b = std::exp(b);
            }
            ha[i] = a;
            hib[i] = 1.0f / (b + 1e-9f);
        }
        float* da = ws.upload(ha);
        float* dib = ws.upload(hib);
        maa::run_sized(c, [&] {
            maa::T4 x = maa::alloc_t(c, B, 1, L, C), y = maa::alloc_t(c, B, 1, L, C);
            maa::launch_nchw_to_nhwc(c, d_x, B, C, L, x.p);
            maa::launch_snake_aa(c, x.p, B, L, C, dib, da, y.p);
            maa::launch_nhwc_to_nchw(c, y.p, B, C, L, d_y, C);
        });
        MAA_HIP(hipStreamSynchronize(c.stream));
    });
}

}  // extern "C"
