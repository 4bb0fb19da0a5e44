// Building blocks shared by the UNet, the VAE and the vocoders: thin wrappers that turn a layer into
// igemm / norm launches on channels-last tensors.
#pragma once
#include "maa_internal.h"

namespace maa {

struct T4 {              // dense channels-last activation [B, H, W, C] (sequences: H = 1)
    float* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    bool split = false;  // true: rows are split32 lines ([32 bf16 hi | 32 bf16 lo] per 32 channels) instead of fp32
    int ld = 0;          // row pitch in floats; 0 = dense (C)
    long long numel() const { return (long long)B * H * W * C; }
    long long rows() const { return (long long)B * H * W; }
};

inline T4 alloc_t(Ctx& ctx, int B, int H, int W, int C) {
    T4 t;
    t.B = B;
    t.H = H;
    t.W = W;
    t.C = C;
    t.p = ctx.ws.alloc_f((size_t)t.numel());
    return t;
}

struct ConvOpt {
    int KH = 1, KW = 1, stride = 1, pad = 0, dil = 1;   // square/isotropic for 2-D; 1-D uses KW only
    int pad_h = -1;                                      // override (VAE downsample pads right/bottom only)
    int up = 0;
    int a_act = 0;
    float a_slope = 0.f;
    const float* rowadd = nullptr;
    int ld_rowadd = 0;
    const float* res = nullptr;   // dense residual with the output's shape
    int act = 0;                  // IGemm::act (4 = leaky-relu with act_slope)
    float act_slope = 0.f;
    float out_scale = 1.f;
    int accumulate = 0;
    int geglu = 0;
    int c_split = 0;              // write the output as split32 lines
    float* c2 = nullptr;          // second output: leaky-relu(c2_slope) of the result as split32 lines (output's shape)
    float c2_slope = 1.f;
};

// out = conv(x1 ++ x2) with packed weight `w`; output spatial size given by (Ho, Wo)
void conv_into(Ctx& ctx, const T4& x1, const T4* x2, const PackedW& w, const ConvOpt& o, T4& out);
// linear over rows of a [rows, K] matrix (any leading layout, row pitch lda)
// Upsample (nearest 2x) + conv3x3 as four 2x2 phase convolutions of the low-resolution source (4 / 9 of the multiplications; bf16
// modes, weights from WeightStore::pack_conv_up2).  false: not applicable here -- the caller runs the 3x3 convolution with ConvOpt::up
bool conv_up2_into(Ctx& ctx, const T4& x, const PackedW& w4, T4& out);
void linear_into(Ctx& ctx, const float* a, int lda, long long rows, int K, const PackedW& w, const float* res,
                 int ldr, float* out, int ldc, int geglu = 0, int a_act = 0, long long a_split_rows = 0,
                 int c_split = 0, int act = 0);
// act: output activation of the igemm epilogue (IGemm::act), applied after bias and residual
// a_split_rows > 0: `a` holds split32 rows (row pitch lda floats);  c_split: write `out` as split32 rows

// softmax(alpha * Q K^T) V for `heads` heads of width dh stored head-major inside rows of q/k/v
//   q: [B, Nq, *] pitch ldq;  k, v: [B, Nk, *] pitch ldk / ldv;  per-head column offset = h * head_stride_{q,k,v}
//   out: [B, Nq, heads*dh] dense
void attention_into(Ctx& ctx, const float* q, int ldq, int hsq, const float* k, int ldk, int hsk, const float* v,
                    int ldv, int hsv, int B, int heads, int dh, int Nq, int Nk, float alpha, float* out, int ldo,
                    int out_split = 0, int causal = 0);
// causal: query i attends to keys 0 .. i only (Nq == Nk)
// out_split: write split32 rows (only where flash_attention_covers(ctx, dh))

// In the bf16 modes a normalisation whose only consumer is a bf16-engine contraction writes bf16 hi/lo planes
// (no per-tile re-splitting of the same activation for every tap and N-tile)
inline bool split_for_gemm(const Ctx& ctx, int C) { return ctx.dtype != 0 && C % 32 == 0; }

// Run a model under its own precision mode (the mode its weights were packed for)
struct PrecisionGuard {
    Ctx& ctx;
    int saved;
    PrecisionGuard(Ctx& c, int mode) : ctx(c), saved(c.dtype) { c.dtype = mode; }
    ~PrecisionGuard() { ctx.dtype = saved; }
};

// Size the workspace with a dry run of `f` (allocations counted, launches skipped), grow the slab if needed
// (never during graph capture: sizes are fixed per shape), then run `f` for real.
template <class F>
void run_sized(Ctx& ctx, F&& f) {
    ctx.ws.reset();
    ctx.ws.dry = true;
    try {
        f();
    } catch (...) {
        ctx.ws.dry = false;
        throw;
    }
    ctx.ws.dry = false;
    const size_t need = ctx.ws.mark_high();
    if (need > ctx.ws.capacity()) ctx.ws.reserve(need + need / 8);
    ctx.ws.reset();
    f();
}

}  // namespace maa
