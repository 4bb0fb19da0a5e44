// Device-resident DDIM loop: S x { build UNet input, UNet forward (2B with CFG), CFG combine + x_{t-1} update }.
//
// Mirrors ldm/models/diffusion/ddim.py:118-166 (ddim_sampling) and :169-225 (p_sample_ddim) of the
// reference: the tools' call pattern (eta = 0, no mask) and, since round 4, the rest of sample()'s signature that is pure
// tensor arithmetic -- mask / x0 blending (:147-150), eta > 0 with the caller's noise (:210-225), the logged intermediates
// (:158-163); score correctors, quantisation, dropout noise and host callbacks stay out:
//   CFG batch order is [uncond ; cond] on x, t and context (:177-199)
//   concat conditioning is cat([x, c], dim=1) (ldm/models/diffusion/ddpm.py:1404-1406)
// Nothing returns to the host inside the loop: per-step scalars (t, a_t, a_prev, sqrt(1-a_t)) are rows of
// device tables copied into fixed slots, so every step launches the same kernels on the same addresses and
// the step can be captured once as a hipGraph and replayed.  The latent and the concat conditioning are staged in the
// context's own slab, so the captured step does not depend on the caller's buffers and is KEPT across sample() calls
// (Ctx::ddim_graph): a later call with the same model, shapes and guidance replays it without capturing again.
#include "models.h"

#include <cmath>
#include <cstdio>
#include <cstring>

// Per-DDIM-step range markers for rocprofv3 --marker-trace (tracing): compiled in by
// `MAA_BUILD_ROCTX=1 python -m audiogpt_amd.build --force` (-DMAA_ROCTX, links libroctx64); the default build has none.
#ifdef MAA_ROCTX
#include <roctracer/roctx.h>
#define MAA_RANGE_PUSH(name) roctxRangePushA(name)
#define MAA_RANGE_POP() roctxRangePop()
#else
#define MAA_RANGE_PUSH(name) ((void)0)
#define MAA_RANGE_POP() ((void)0)
#endif

namespace maa {

void ddim_sample(Ctx& ctx, UNet& unet, const maa_ddim_args& a, float* d_x) {
    MAA_CHECK(a.S > 0 && a.B > 0, "ddim: empty problem");
    MAA_CHECK(!a.d_uncond || a.d_cond, "ddim: unconditional conditioning given without conditioning");
    MAA_CHECK(!a.d_cond || a.L > 0, "ddim: conditioning needs its token count L");
    MAA_CHECK(!a.d_concat || a.Cc > 0, "ddim: concat conditioning needs its channel count");
    const bool concat = a.d_concat != nullptr;
    const bool cfg = !concat && a.d_uncond != nullptr && a.scale != 1.0f;
    const int nB = cfg ? 2 * a.B : a.B;
    const int Cin = concat ? a.C + a.Cc : a.C;
    const long long per = (long long)a.C * a.H * a.W;           // latent elements per sample
    const long long per_in = (long long)Cin * a.H * a.W;
    MAA_CHECK(unet.config().in_channels == Cin, "ddim: UNet in_channels does not match latent (+concat) channels");

    // ---- device state of the loop, in one slab the context keeps across calls: tables (one row per DDIM index), the
    // device step index, the step's timestep / coefficient slots, UNet input and output
    const bool masked = a.d_mask != nullptr;
    MAA_CHECK(!masked || (a.d_x0 && a.d_noise_q && a.h_sqrt_ac && a.h_sqrt_1mac), "ddim: mask needs x0, its noise and the q_sample tables");
    MAA_CHECK(!a.h_sigmas || a.d_noise_p, "ddim: eta > 0 needs the steps' noise");
    const bool logging = a.n_log > 0;
    MAA_CHECK(!logging || (a.d_log_x && a.d_log_x0 && a.log_every_t > 0), "ddim: intermediates need their buffers and log_every_t");
    std::vector<float> h_tab((size_t)a.S * 9);
    int n_logged = 0;
    for (int v = 0; v < a.S; ++v) {                    // visiting order: index S-1 first (ddim.py:143-145)
        const int i = a.S - 1 - v;
        h_tab[i] = (float)a.h_timesteps[i];
        float* cf = &h_tab[(size_t)a.S + (size_t)i * 8];
        cf[0] = a.h_alphas[i];
        cf[1] = a.h_alphas_prev[i];
        cf[2] = a.h_sigmas ? a.h_sigmas[i] : 0.f;      // eta = 0: no noise term
        cf[3] = std::sqrt(1.0f - a.h_alphas[i]);       // ddim.py:52 (fp32 sqrt of fp32 1-a)
        cf[4] = masked ? a.h_sqrt_ac[i] : 0.f;
        cf[5] = masked ? a.h_sqrt_1mac[i] : 0.f;
        const bool logged = logging && (i % a.log_every_t == 0 || i == a.S - 1);      // ddim.py:161
        cf[6] = logged ? (float)n_logged++ : -1.f;
        cf[7] = (float)i;
    }
    MAA_CHECK(!logging || n_logged == a.n_log, "ddim: n_log does not match log_every_t");
    auto up = [](size_t n) { return (n + 63) / 64 * 64; };      // floats, 256-byte aligned pieces
    const size_t n_cc = concat ? (size_t)a.B * (per_in - per) : 0;
    // the ResBlocks' time-embedding rows of all S steps, computed once per call (every sample of a step shares t; the I2A variant
    // adds the sample's context to the embedding and keeps the per-forward computation): six launches leave every step
    const bool emb_hoist = !unet.config().add_context_to_emb;
    const size_t emb_w = emb_hoist ? (size_t)unet.emb_width() : 0;
    const size_t o_tab = 0, o_step = o_tab + up(h_tab.size()), o_t = o_step + 64, o_coef = o_t + up(nB),
                 o_xin = o_coef + 64, o_eps = o_xin + up((size_t)nB * per_in), o_x = o_eps + up((size_t)nB * per),
                 o_cc = o_x + up((size_t)a.B * per), o_embt = o_cc + up(n_cc), o_emb = o_embt + up((size_t)a.S * emb_w),
                 total = o_emb + up(emb_w);
    float* slab = static_cast<float*>(ctx.sampler_scratch.get(total * sizeof(float), ctx.stream));
    float *tab_t = slab + o_tab, *tab_coef = slab + o_tab + a.S, *cur_t = slab + o_t, *cur_coef = slab + o_coef,
          *xin = slab + o_xin, *eps = slab + o_eps, *xs = slab + o_x, *ccs = slab + o_cc, *emb_tab = slab + o_embt,
          *cur_emb = slab + o_emb;
    // the trajectory runs on the slab's copy of the latent (and of the concat conditioning)
    MAA_HIP(hipMemcpyAsync(xs, d_x, (size_t)a.B * per * 4, hipMemcpyDeviceToDevice, ctx.stream));
    if (concat) MAA_HIP(hipMemcpyAsync(ccs, a.d_concat, n_cc * 4, hipMemcpyDeviceToDevice, ctx.stream));
    int* d_step = reinterpret_cast<int*>(slab + o_step);
    const int h_step = a.S - 1;                        // ddim.py:143-145: flipped timesteps, index = total - i - 1
    MAA_HIP(hipMemcpyAsync(slab + o_tab, h_tab.data(), h_tab.size() * 4, hipMemcpyHostToDevice, ctx.stream));
    MAA_HIP(hipMemcpyAsync(d_step, &h_step, 4, hipMemcpyHostToDevice, ctx.stream));

    if (emb_hoist) unet.emb_table(ctx, tab_t, a.S, emb_tab);      // (tab_t: the S timesteps as floats, uploaded above)

    // ---- conditioning: constant over the trajectory -> project K/V once
    if (!concat && a.d_cond) {
        if (cfg)
            unet.set_context_cfg(ctx, a.d_uncond, a.d_cond, a.B, a.L);
        else
            unet.set_context(ctx, a.d_cond, nB, a.L);
    }

    // Classifier-free guidance: the reference evaluates the model once on cat([x] * 2) (ddim.py:177-199); the two halves are
    // independent until the combine, and one batch of 8 prompts leaves much of the chip idle, so the step forks
    // after the UNet input is built -- the unconditional half on the context's stream, the conditional half on the second lane's
    // stream and workspace -- and joins before the update kernel.  Captured, the halves are two branches of the step graph.
    // Every kernel is batch-invariant bit for bit, so the result equals the one-stream form's.
    const bool two_lanes = cfg && ctx.split_cfg();
    Ctx* lane2 = two_lanes ? &side_lane(ctx) : nullptr;
    if (lane2 && ctx.prof && !lane2->prof) {
        lane2->prof = new Profiler;
        lane2->prof->detail = ctx.prof->detail;
    }

    // a guided step's halves are the same tensor up to the first cross-attention: computed once (MAA_CFG_SHARED=0: twice)
    const bool share = cfg && emb_hoist && ctx.tune.cfg_shared;
    // one step: identical launches on identical addresses whatever the step (the index lives on the device)
    auto step_body = [&]() {
        launch_ddim_prepare(ctx, xs, concat ? ccs : nullptr, a.B, nB, per, per_in - per, tab_t, tab_coef, d_step, xin,
                            cur_t, cur_coef, a.d_mask, a.d_x0, a.d_noise_q, a.S, emb_hoist ? emb_tab : nullptr, (int)emb_w, cur_emb);
        if (lane2) {
            MAA_HIP(hipEventRecord(ctx.ev_fork, ctx.stream));
            MAA_HIP(hipStreamWaitEvent(lane2->stream, ctx.ev_fork, 0));
            // (the conditional lane starts from the unconditional lane's layers before the first cross-attention: unet.cpp)
            unet.forward(ctx, xin, cur_t, unet.context_ptr, a.B, a.H, a.W, eps, emb_hoist ? cur_emb : nullptr, 0, share ? 2 : 0);
            unet.forward(*lane2, xin + (size_t)a.B * per_in, cur_t + a.B, unet.context_ptr, a.B, a.H, a.W, eps + (size_t)a.B * per,
                         emb_hoist ? cur_emb : nullptr, a.B, share ? 3 : 0);
            MAA_HIP(hipEventRecord(ctx.ev_join, lane2->stream));
            MAA_HIP(hipStreamWaitEvent(ctx.stream, ctx.ev_join, 0));
        } else
            // (one stream: the halves of cat([x] * 2) share every layer before the first cross-attention -- unet.cpp `dup`)
            unet.forward(ctx, xin, cur_t, unet.context_ptr, nB, a.H, a.W, eps, emb_hoist ? cur_emb : nullptr, -1, share ? 1 : 0);
        launch_ddim_step(ctx, xin, per, per_in, eps, cfg ? eps + a.B * per : nullptr, a.scale, cur_coef, (long long)a.B * per, xs,
                         a.h_sigmas ? a.d_noise_p : nullptr, a.temperature, a.S, logging ? a.d_log_x : nullptr,
                         logging ? a.d_log_x0 : nullptr, d_step);
    };

    // Everything a captured step depends on besides the device-side state it reads: the model and its own buffers, the
    // shapes, the guidance scale, the slab (every slot's offset is a function of the numbers listed) and the workspace.
    // (The workspace base / capacity go in AFTER the first eager step, which may grow it.)
    auto make_key = [&]() {
        std::vector<unsigned long long> k;
        unet.graph_key(k);
        unsigned scale_bits, temp_bits;
        static_assert(sizeof(scale_bits) == sizeof(a.scale), "float bits");
        std::memcpy(&scale_bits, &a.scale, 4);
        std::memcpy(&temp_bits, &a.temperature, 4);
        k.push_back(temp_bits);
        for (unsigned long long v : {(unsigned long long)a.S, (unsigned long long)a.B, (unsigned long long)a.C, (unsigned long long)a.H,
                                     (unsigned long long)a.W, (unsigned long long)a.Cc, (unsigned long long)a.L,
                                     (unsigned long long)cfg, (unsigned long long)concat, (unsigned long long)scale_bits,
                                     (unsigned long long)ctx.dtype, (unsigned long long)reinterpret_cast<uintptr_t>(slab),
                                     // the caller's buffers the step's launches read or write besides the slab
                                     (unsigned long long)reinterpret_cast<uintptr_t>(a.d_mask),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(a.d_x0),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(a.d_noise_q),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(a.h_sigmas ? a.d_noise_p : nullptr),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(logging ? a.d_log_x : nullptr),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(logging ? a.d_log_x0 : nullptr),
                                     (unsigned long long)(a.h_sigmas ? 1 : 0),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(ctx.stream),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(ctx.ws.base()),
                                     (unsigned long long)ctx.ws.capacity(),
                                     // the second lane of a CFG step: its stream and workspace are part of the captured step
                                     (unsigned long long)reinterpret_cast<uintptr_t>(lane2 ? lane2->stream : nullptr),
                                     (unsigned long long)reinterpret_cast<uintptr_t>(lane2 ? lane2->ws.base() : nullptr),
                                     (unsigned long long)(lane2 ? lane2->ws.capacity() : 0)})
            k.push_back(v);
        return k;
    };

    StepGraph& sg = ctx.ddim_graph;
    int first = 0;
    if (a.use_graph && sg.exec && sg.key == make_key()) {
        // same step as the last call's: replay from the first step on (the workspace the graph was captured over is still
        // this context's, at the same address and size)
    } else if (a.use_graph) {
        sg.clear();
        step_body();                      // first step eager: sizes the workspace before any capture
        first = 1;
        if (a.S > 1) {
            MAA_HIP(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeRelaxed));
            try {
                step_body();
            } catch (...) {
                hipGraph_t dead = nullptr;
                (void)hipStreamEndCapture(ctx.stream, &dead);
                if (dead) (void)hipGraphDestroy(dead);
                throw;
            }
            MAA_HIP(hipStreamEndCapture(ctx.stream, &sg.graph));
            MAA_HIP(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0));
            sg.key = make_key();
        }
    }
    for (int i = first; i < a.S; ++i) {
#ifdef MAA_ROCTX
        char range[48];
        std::snprintf(range, sizeof(range), "ddim_step %d/%d t=%d", i + 1, a.S, (int)a.h_timesteps[a.S - 1 - i]);
#endif
        MAA_RANGE_PUSH(range);
        if (a.use_graph)
            MAA_HIP(hipGraphLaunch(sg.exec, ctx.stream));
        else
            step_body();
        MAA_RANGE_POP();
    }
    MAA_HIP(hipMemcpyAsync(d_x, xs, (size_t)a.B * per * 4, hipMemcpyDeviceToDevice, ctx.stream));
    MAA_HIP(hipStreamSynchronize(ctx.stream));   // the host tables go out of scope; the call returns a finished latent
}

}  // namespace maa
