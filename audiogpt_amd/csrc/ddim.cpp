// Device-resident DDIM loop: S x { build UNet input, UNet forward (2B with CFG), CFG combine + x_{t-1} update }.
//
// Mirrors ldm/models/diffusion/ddim.py:118-166 (ddim_sampling) and :169-225 (p_sample_ddim) of the
// reference for the tools' call pattern (eta = 0, no mask / score corrector / quantisation):
//   CFG batch order is [uncond ; cond] on x, t and context (:177-199)
//   concat conditioning is cat([x, c], dim=1) (ldm/models/diffusion/ddpm.py:1404-1406)
// Nothing returns to the host inside the loop: per-step scalars (t, a_t, a_prev, sqrt(1-a_t)) are rows of
// device tables copied into fixed slots, so every step launches the same kernels on the same addresses and
// the step can be captured once as a hipGraph and replayed.
#include "models.h"

#include <cmath>

namespace maa {

namespace {
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { MAA_HIP(hipMalloc(&p, bytes ? bytes : 4)); }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    float* f() const { return static_cast<float*>(p); }
};
}  // namespace

void ddim_sample(Ctx& ctx, UNet& unet, const maa_ddim_args& a, float* d_x) {
    MAA_CHECK(a.S > 0 && a.B > 0, "ddim: empty problem");
    const bool concat = a.d_concat != nullptr;
    const bool cfg = !concat && a.d_uncond != nullptr && a.scale != 1.0f;
    const int nB = cfg ? 2 * a.B : a.B;
    const int Cin = concat ? a.C + a.Cc : a.C;
    const long long per = (long long)a.C * a.H * a.W;           // latent elements per sample
    const long long per_in = (long long)Cin * a.H * a.W;
    MAA_CHECK(unet.config().in_channels == Cin, "ddim: UNet in_channels does not match latent (+concat) channels");

    // ---- device tables: one row per DDIM index
    std::vector<float> h_t((size_t)a.S * nB), h_coef((size_t)a.S * 4);
    for (int i = 0; i < a.S; ++i) {
        for (int b = 0; b < nB; ++b) h_t[(size_t)i * nB + b] = (float)a.h_timesteps[i];
        h_coef[(size_t)i * 4 + 0] = a.h_alphas[i];
        h_coef[(size_t)i * 4 + 1] = a.h_alphas_prev[i];
        h_coef[(size_t)i * 4 + 2] = 0.f;                                   // eta = 0
        h_coef[(size_t)i * 4 + 3] = std::sqrt(1.0f - a.h_alphas[i]);       // ddim.py:52 (fp32 sqrt of fp32 1-a)
    }
    DevBuf tab_t(h_t.size() * 4), tab_coef(h_coef.size() * 4), cur_t((size_t)nB * 4), cur_coef(16);
    DevBuf xin((size_t)nB * per_in * 4), eps((size_t)nB * per * 4), ctxbuf(cfg ? (size_t)nB * a.L * unet.config().context_dim * 4 : 4);
    MAA_HIP(hipMemcpyAsync(tab_t.p, h_t.data(), h_t.size() * 4, hipMemcpyHostToDevice, ctx.stream));
    MAA_HIP(hipMemcpyAsync(tab_coef.p, h_coef.data(), h_coef.size() * 4, hipMemcpyHostToDevice, ctx.stream));

    // ---- conditioning: constant over the trajectory -> project K/V once
    if (!concat && a.d_cond) {
        const size_t cbytes = (size_t)a.B * a.L * unet.config().context_dim * 4;
        if (cfg) {
            MAA_HIP(hipMemcpyAsync(ctxbuf.p, a.d_uncond, cbytes, hipMemcpyDeviceToDevice, ctx.stream));
            MAA_HIP(hipMemcpyAsync(static_cast<char*>(ctxbuf.p) + cbytes, a.d_cond, cbytes, hipMemcpyDeviceToDevice,
                                   ctx.stream));
            unet.set_context(ctx, ctxbuf.f(), nB, a.L);
        } else {
            unet.set_context(ctx, a.d_cond, nB, a.L);
        }
    }

    auto step_body = [&]() {
        // UNet input
        if (cfg) {
            MAA_HIP(hipMemcpyAsync(xin.p, d_x, (size_t)a.B * per * 4, hipMemcpyDeviceToDevice, ctx.stream));
            MAA_HIP(hipMemcpyAsync(xin.f() + a.B * per, d_x, (size_t)a.B * per * 4, hipMemcpyDeviceToDevice, ctx.stream));
        } else if (concat) {
            MAA_HIP(hipMemcpy2DAsync(xin.p, (size_t)per_in * 4, d_x, (size_t)per * 4, (size_t)per * 4, a.B,
                                     hipMemcpyDeviceToDevice, ctx.stream));
            MAA_HIP(hipMemcpy2DAsync(xin.f() + per, (size_t)per_in * 4, a.d_concat, (size_t)(per_in - per) * 4,
                                     (size_t)(per_in - per) * 4, a.B, hipMemcpyDeviceToDevice, ctx.stream));
        } else {
            MAA_HIP(hipMemcpyAsync(xin.p, d_x, (size_t)a.B * per * 4, hipMemcpyDeviceToDevice, ctx.stream));
        }
        unet.forward(ctx, xin.f(), cur_t.f(), unet.context_ptr, nB, a.H, a.W, eps.f());
        launch_ddim_update(ctx, d_x, eps.f(), cfg ? eps.f() + a.B * per : nullptr, a.scale, cur_coef.f(),
                           (long long)a.B * per, d_x, nullptr);
    };

    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    try {
        for (int i = 0; i < a.S; ++i) {
            const int index = a.S - 1 - i;       // ddim.py:143-145: flipped timesteps, index = total - i - 1
            MAA_HIP(hipMemcpyAsync(cur_t.p, tab_t.f() + (size_t)index * nB, (size_t)nB * 4, hipMemcpyDeviceToDevice,
                                   ctx.stream));
            MAA_HIP(hipMemcpyAsync(cur_coef.p, tab_coef.f() + (size_t)index * 4, 16, hipMemcpyDeviceToDevice, ctx.stream));
            if (!a.use_graph || i == 0) {
                step_body();                      // first step eager: sizes the workspace before any capture
            } else {
                if (!exec) {
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
        }
        MAA_HIP(hipStreamSynchronize(ctx.stream));   // host tables and scratch buffers go out of scope
    } catch (...) {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
}

}  // namespace maa
