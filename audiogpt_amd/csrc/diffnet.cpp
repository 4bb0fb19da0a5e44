// DiffSinger's denoiser (DiffNet) and the PLMS sampling loop on the device (the T2S tool's diffusion hot
// loop, audio-chatgpt.py:298-339).
//
// Mirrors NeuralSeq/modules/diff/net.py:58-130 (ResidualBlock, DiffNet.forward) and
// NeuralSeq/modules/diff/shallow_diffusion_tts.py:166-201, 262-269 (p_sample_plms and its loop with pndm_speedup).
// Layout: channels-last sequences [B, T, C]; a Conv1d(k = 1) is a plain GEMM over the B*T rows, the dilated k = 3 conv an
// implicit GEMM.  Differences from a literal translation (same maths):
//   * the 20 per-layer `diffusion_projection` Linears of the step embedding run as ONE GEMM per evaluation
//   * the 20 per-layer `conditioner_projection` convs of the (step-invariant) conditioning run ONCE per sample() call,
//     as one GEMM, and enter each layer's dilated conv as its residual epilogue
//   * relu, bias, the conditioner add and 1/sqrt(n_layers) are igemm epilogues; the gate, the residual/skip update and
//     the next layer's `x + step` are two elementwise kernels per layer
//   * the sampling loop keeps t, the noise-history ring and its fill count on the device, so every step after the
//     first is the same launch sequence: captured once as a hipGraph and replayed
#include "models.h"

#include <cmath>

namespace maa {

void launch_ds_pos_emb(const Ctx& ctx, const float* t, int B, int dim, float* out);
void launch_ds_mish(const Ctx& ctx, const float* x, long long n, float* out);
void launch_ds_add_step(const Ctx& ctx, const float* x, const float* step, int ld_step, long long rows, int T, int C,
                        float* out);
void launch_ds_gate(const Ctx& ctx, const float* y, long long rows, int C, float* z);
void launch_ds_residual(const Ctx& ctx, const float* y2, long long rows, int T, int C, float* x, float* skip, int first,
                        const float* next_step, int ld_step, float* xin);
void launch_ds_plms(const Ctx& ctx, float* x, const float* e, const float* e_prev, float* hist, long long n,
                    const float* ac, int interval, int* st, int mode, float* x_out);
void launch_ds_plms_advance(const Ctx& ctx, int* st, int interval, float* t_slot, int B);
void launch_ds_fill(const Ctx& ctx, float* p, int n, float v);

struct DiffNet::Impl {
    maa_diffnet_config cfg;
    int precision = 0;
    WeightStore ws;
    explicit Impl(int prec) : precision(prec), ws(prec != 0) {}
    PackedW in_proj, mlp0, mlp2, step_all, cond_all, skip_proj, out_proj;
    std::vector<PackedW> dilated, out_layer;
    DevSlab cond_cache;       // conditioner projections of all layers for the current conditioning: [B*T, L * 2C]
    const float* cond_src = nullptr;
    int cond_B = 0, cond_T = 0;
    DevSlab loop;             // PLMS loop state

    void build(const StateDict& sd) {
        const int L = cfg.residual_layers;
        in_proj = ws.pack_conv(sd, "input_projection.weight", "input_projection.bias", 1, 1);
        mlp0 = ws.pack_conv(sd, "mlp.0.weight", "mlp.0.bias", 1, 1);
        mlp2 = ws.pack_conv(sd, "mlp.2.weight", "mlp.2.bias", 1, 1);
        std::vector<std::string> sw, sb, cw, cb;
        for (int i = 0; i < L; ++i) {
            const std::string p = "residual_layers." + std::to_string(i) + ".";
            dilated.push_back(ws.pack_conv(sd, p + "dilated_conv.weight", p + "dilated_conv.bias", 1, 3));
            out_layer.push_back(ws.pack_conv(sd, p + "output_projection.weight", p + "output_projection.bias", 1, 1));
            sw.push_back(p + "diffusion_projection.weight");
            sb.push_back(p + "diffusion_projection.bias");
            cw.push_back(p + "conditioner_projection.weight");
            cb.push_back(p + "conditioner_projection.bias");
        }
        step_all = ws.pack_concat(sd, sw, sb);
        cond_all = ws.pack_concat(sd, cw, cb);
        MAA_CHECK(step_all.N == L * cfg.residual_channels && cond_all.N == L * 2 * cfg.residual_channels, "DiffNet packing");
        skip_proj = ws.pack_conv(sd, "skip_projection.weight", "skip_projection.bias", 1, 1);
        out_proj = ws.pack_conv(sd, "output_projection.weight", "output_projection.bias", 1, 1);
    }

    // conditioner projections of every layer: cond [B, H, T] (reference layout) -> cache [B*T, L*2C]
    void set_cond(Ctx& ctx, const float* cond, int B, int T) {
        const size_t rows = (size_t)B * T;
        float* cache = static_cast<float*>(cond_cache.get(rows * cond_all.Npad * sizeof(float), ctx.stream));
        run_sized(ctx, [&] {
            float* cl = ctx.ws.alloc_f(rows * cfg.hidden_size);
            launch_nchw_to_nhwc(ctx, cond, B, cfg.hidden_size, T, cl);
            linear_into(ctx, cl, cfg.hidden_size, (long long)rows, cfg.hidden_size, cond_all, nullptr, 0, cache, cond_all.Npad);
        });
        cond_src = cond;
        cond_B = B;
        cond_T = T;
    }

    // spec [B, 1, M, T], t [B] (float), conditioning set by set_cond -> eps [B, 1, M, T]
    void forward(Ctx& ctx, const float* spec, const float* t, int B, int T, float* out) {
        MAA_CHECK(ctx.ws.dry || (cond_B == B && cond_T == T), "DiffNet: set the conditioning for this (B, T) first");
        const int C = cfg.residual_channels, L = cfg.residual_layers, M = cfg.in_dims;
        const long long rows = (long long)B * T;
        const float* cache = static_cast<const float*>(cond_cache.p);
        // x = relu(input_projection(spec))
        float* sp = ctx.ws.alloc_f((size_t)rows * M);
        launch_nchw_to_nhwc(ctx, spec, B, M, T, sp);
        float* x = ctx.ws.alloc_f((size_t)rows * C);
        {
            T4 a, o;
            a.B = o.B = B;
            a.H = o.H = 1;
            a.W = o.W = T;
            a.C = M;
            o.C = C;
            a.p = sp;
            o.p = x;
            ConvOpt co;
            co.act = 2;
            conv_into(ctx, a, nullptr, in_proj, co, o);
        }
        // diffusion-step embedding -> mlp -> all layers' diffusion_projection
        float* pe = ctx.ws.alloc_f((size_t)B * C);
        launch_ds_pos_emb(ctx, t, B, C, pe);
        float* h = ctx.ws.alloc_f((size_t)B * 4 * C);
        linear_into(ctx, pe, C, B, C, mlp0, nullptr, 0, h, 4 * C);
        launch_ds_mish(ctx, h, (long long)B * 4 * C, h);
        float* d = ctx.ws.alloc_f((size_t)B * C);
        linear_into(ctx, h, 4 * C, B, 4 * C, mlp2, nullptr, 0, d, C);
        float* steps = ctx.ws.alloc_f((size_t)B * step_all.Npad);
        linear_into(ctx, d, C, B, C, step_all, nullptr, 0, steps, step_all.Npad);

        float* xin = ctx.ws.alloc_f((size_t)rows * C);
        float* y = ctx.ws.alloc_f((size_t)rows * 2 * C);
        float* z = ctx.ws.alloc_f((size_t)rows * C);
        float* skip = ctx.ws.alloc_f((size_t)rows * C);
        launch_ds_add_step(ctx, x, steps, step_all.Npad, rows, T, C, xin);
        for (int i = 0; i < L; ++i) {
            const int dil = 1 << (i % cfg.dilation_cycle_length);
            T4 a, o;
            a.B = o.B = B;
            a.H = o.H = 1;
            a.W = o.W = T;
            a.C = C;
            o.C = 2 * C;
            a.p = xin;
            o.p = y;
            // y = dilated_conv(x + step) + conditioner_projection(cond)     (net.py:71-74)
            IGemm p;
            p.a1 = xin;
            p.lda1 = C;
            p.C1 = C;
            p.Hin = p.Hout = 1;
            p.Win = p.Wout = T;
            p.KH = 1;
            p.KW = 3;
            p.dw = dil;
            p.pw = dil;
            p.b = dilated[i].w;
            p.ldb = dilated[i].ld;
            p.b_nk = dilated[i].nk;
            p.b_split = dilated[i].split;
            p.M = (int)rows;
            p.K = 3 * C;
            p.N = 2 * C;
            p.bias = dilated[i].bias;
            p.res = cache + (size_t)i * 2 * C;
            p.ldr = cond_all.Npad;
            p.c = y;
            p.ldc = 2 * C;
            launch_igemm(ctx, p);
            launch_ds_gate(ctx, y, rows, C, z);
            // y2 = output_projection(z);  x = (x + y2[:C]) / sqrt(2);  skip += y2[C:]     (:78-81)
            linear_into(ctx, z, C, rows, C, out_layer[i], nullptr, 0, y, 2 * C);
            const bool last = i + 1 == L;
            launch_ds_residual(ctx, y, rows, T, C, x, skip, i == 0, last ? nullptr : steps + (size_t)(i + 1) * C,
                               step_all.Npad, xin);
        }
        // x = relu(skip_projection(sum(skip) / sqrt(L)));  out = output_projection(x)      (net.py:125-129)
        {
            IGemm p;
            p.a1 = skip;
            p.lda1 = C;
            p.C1 = C;
            p.M = (int)rows;
            p.K = C;
            p.N = C;
            p.b = skip_proj.w;
            p.ldb = skip_proj.ld;
            p.b_nk = skip_proj.nk;
            p.b_split = skip_proj.split;
            p.bias = skip_proj.bias;
            p.alpha = 1.0f / std::sqrt((float)L);
            p.act = 2;
            p.c = z;
            p.ldc = C;
            launch_igemm(ctx, p);
        }
        float* o80 = ctx.ws.alloc_f((size_t)rows * M);
        linear_into(ctx, z, C, rows, C, out_proj, nullptr, 0, o80, M);
        launch_nhwc_to_nchw(ctx, o80, B, M, T, out, M);
    }
};

DiffNet::DiffNet(const maa_diffnet_config& cfg, const StateDict& sd, int precision) : impl_(new Impl(precision)) {
    impl_->cfg = cfg;
    impl_->build(sd);
}
DiffNet::~DiffNet() { delete impl_; }
const maa_diffnet_config& DiffNet::config() const { return impl_->cfg; }

void DiffNet::forward(Ctx& ctx, const float* spec, const float* t, const float* cond, int B, int T, float* out) {
    Impl& m = *impl_;
    PrecisionGuard pg(ctx, m.precision);
    m.set_cond(ctx, cond, B, T);
    run_sized(ctx, [&] { m.forward(ctx, spec, t, B, T, out); });
}

void DiffNet::plms_sample(Ctx& ctx, const maa_plms_args& a, float* d_x) {
    Impl& m = *impl_;
    MAA_CHECK(a.B > 0 && a.T > 0 && a.K_step > 0 && a.interval > 0 && a.timesteps >= a.K_step && a.h_alphas_cumprod && a.d_cond,
              "bad plms arguments");
    MAA_CHECK(a.B <= 256, "plms: at most 256 samples per call (the step-advance kernel is one workgroup)");
    PrecisionGuard pg(ctx, m.precision);
    const int B = a.B, T = a.T, M = m.cfg.in_dims;
    const long long n = (long long)B * M * T;
    m.set_cond(ctx, a.d_cond, B, T);
    // loop state: alphas_cumprod table, {t, history count, ring head}, timestep slot, eps, eps_prev / x_pred, history ring
    auto up = [](size_t k) { return (k + 63) / 64 * 64; };
    const size_t o_ac = 0, o_st = o_ac + up(a.timesteps), o_t = o_st + 64, o_e = o_t + up(B), o_e2 = o_e + up(n),
                 o_xp = o_e2 + up(n), o_h = o_xp + up(n), total = o_h + 3 * up(n);
    float* slab = static_cast<float*>(m.loop.get(total * sizeof(float), ctx.stream));
    float *ac = slab + o_ac, *t_slot = slab + o_t, *e = slab + o_e, *e2 = slab + o_e2, *xp = slab + o_xp, *hist = slab + o_h;
    int* st = reinterpret_cast<int*>(slab + o_st);
    // first t of reversed(range(0, K_step, interval))
    const int t0 = ((a.K_step - 1) / a.interval) * a.interval;
    const int nsteps = t0 / a.interval + 1;
    const int h_st[3] = {t0, 0, 0};
    MAA_HIP(hipMemcpyAsync(ac, a.h_alphas_cumprod, (size_t)a.timesteps * 4, hipMemcpyHostToDevice, ctx.stream));
    MAA_HIP(hipMemcpyAsync(st, h_st, sizeof(h_st), hipMemcpyHostToDevice, ctx.stream));
    const long long hstride = (long long)up(n);
    (void)hstride;
    auto denoise = [&](const float* x, float* out) { run_sized(ctx, [&] { m.forward(ctx, x, t_slot, B, T, out); }); };

    // ---- first step (:184-187): predictor with e, second evaluation at max(t - interval, 0), corrector with the mean
    launch_ds_fill(ctx, t_slot, B, (float)t0);
    denoise(d_x, e);
    launch_ds_plms(ctx, d_x, e, nullptr, hist, n, ac, a.interval, st, 1, xp);
    launch_ds_fill(ctx, t_slot, B, (float)std::max(t0 - a.interval, 0));
    denoise(xp, e2);
    launch_ds_plms(ctx, d_x, e, e2, hist, n, ac, a.interval, st, 2, nullptr);
    launch_ds_plms_advance(ctx, st, a.interval, t_slot, B);

    // ---- remaining steps: identical launches on identical addresses -> one hipGraph, replayed
    auto step_body = [&]() {
        m.forward(ctx, d_x, t_slot, B, T, e);          // (workspace already sized by the eager evaluations above)
        launch_ds_plms(ctx, d_x, e, nullptr, hist, n, ac, a.interval, st, 0, nullptr);
        launch_ds_plms_advance(ctx, st, a.interval, t_slot, B);
    };
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    try {
        for (int i = 1; i < nsteps; ++i) {
            if (!a.use_graph) {
                ctx.ws.reset();
                step_body();
                continue;
            }
            if (!exec) {
                ctx.ws.reset();
                MAA_HIP(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeRelaxed));
                try {
                    step_body();
                } catch (...) {
                    hipGraph_t dead = nullptr;
                    (void)hipStreamEndCapture(ctx.stream, &dead);
                    if (dead) (void)hipGraphDestroy(dead);
                    throw;
                }
                MAA_HIP(hipStreamEndCapture(ctx.stream, &graph));
                MAA_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            }
            MAA_HIP(hipGraphLaunch(exec, ctx.stream));
        }
        MAA_HIP(hipStreamSynchronize(ctx.stream));
    } catch (...) {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
}

}  // namespace maa
