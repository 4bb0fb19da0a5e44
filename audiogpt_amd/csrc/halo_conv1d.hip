// Dilated "same" Conv1d for the narrow stages of the HiFi-GAN / BigVGAN MRF resblocks (C = 32 or 64 channels in and out),
// bf16x3 arithmetic: the input tile is staged ONCE in LDS and every tap reads it at a row offset.
//
// Replaces, at those widths, Conv1d(C, C, k, dilation = d, padding = (k d - d) / 2) with its preceding leaky-ReLU inside
// ResBlock1 (NeuralSeq/modules/hifigan/hifigan.py:30-67) and the convs of AMPBlock1 (vocoder/bigvgan/models.py:30-81).
//
// Why.  At 22.05 kHz config-3 scale these layers are [64 x 262144, 32] and [64 x 131072, 64] tensors: 2.1 GB each way.  As a
// generic implicit GEMM a workgroup fetches its 256 x 32-channel A rows again for each of the k taps (k = 3, 7, 11): 36 KB
// through the texture path per 384 MFMA cycles = 96 B/clk/CU, above what that path moves (~64 B/clk/CU), so the layer runs
// at 48 TFLOP/s (1.3 TB/s of HBM-side bytes) -- bound by L1/L2 re-reads, not by HBM and not by the MFMAs.  Per output
// element the layer needs 2 k C multiply-adds (192 .. 1408 flops at C = 32 / 64) against 12 bytes of HBM (input, residual,
// output): at 8 TB/s the HBM bound is 16x (C = 32, k = 3) .. 1.1x (C = 64, k = 11) below the bf16x3 MFMA bound, i.e. the
// layer should be HBM-bound.  Staging the tile once makes the L2->CU bytes equal to the HBM bytes.
//
// One workgroup (4 waves) per tile of TL = 256 (C = 32) or 128 (C = 64) output positions of one sample; persistent over tiles.
//   * A image: rows l0 - H .. l0 + TL + H (H = d (k-1) / 2 <= 25) read as fp32 (coalesced float4 x 2 per lane), leaky-ReLU
//     applied, split into bf16 hi / lo and written as split32 lines (ds_write_b128, slot XOR-swizzled by (row >> 1) & 7)
//     -- rows outside [0, L) are zeros (the conv's zero padding; leaky(0) = 0)
//   * B: the packed split32 weights [C][k C] stream through a 4-stage LDS ring by LDS-DMA, chunk c = (tap, 32-channel block)
//   * wave w owns positions 64 w .. 64 w + 63 (two 32-row MFMA blocks) x all C outputs; for chunk (tap t, block cb) its A
//     fragments are rows i + t d of line cb of the image -- no data moves, only the row index
//   * products per accumulator: lo.hi, hi.lo, hi.hi per 16-deep k-step, K ordered (tap, channel) -- exactly the generic
//     engines' arithmetic, so results are bit-identical to them (tests)
//   * epilogue shared with the implicit-GEMM engines (bias, residual, out_scale, accumulate)
#include "igemm_epilogue.h"

#include <cstdlib>
#include <type_traits>

namespace maa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HMAX = 32;           // halo rows reserved on each side (needs d (k-1) / 2 <= HMAX)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);      // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pk_bf16(a, b);
    lo = pk_bf16(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}

struct HaloArgs {
    IGemm g;            // epilogue fields (bias, res, ldr, out_scale, accumulate, c, ldc, N, alpha), b / ldb (weights), zeros
    const float* x;     // [B, L, C] fp32
    int B, L, k, dil;
    float slope;        // leaky-ReLU slope of the prologue (1 = none)
    int tiles_per_sample, tiles;
};

// TL = output positions per tile (4 waves x MI 32-row blocks); NSB = weight ring stages
template <int C, int TL, int NSB>
__global__ __launch_bounds__(256) void halo_conv1d_kernel(const HaloArgs a) {
    constexpr int AROWS = TL + 2 * HMAX;
    constexpr int CB = C / 32;                 // 128-byte lines per row = 32-channel blocks
    constexpr int NI = C / 32, MI = TL / 128;
    static_assert(TL == 128 || TL == 256, "tile length");
    constexpr int A_BYTES = AROWS * CB * 128;
    constexpr int BSTAGE = C * 128;            // one weight chunk: C rows x 128 B
    constexpr int IPW = C / 32;                // LDS-DMA copies (8 rows x 128 B) per wave and chunk: C / 8 / 4 waves
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [AROWS][CB][128] | [NSB][C][128]
    char* bring = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;
    const int H = a.dil * (a.k - 1) / 2;
    const int nchunks = a.k * CB;

    // weight copies: copy q covers weight rows 8q .. 8q+7; this wave issues q = j 4 + wid (j < IPW)
    const int r8 = lane >> 3;
    const int bslot = (((lane & 7) ^ (((wid & 1) << 2) + (r8 >> 1))) << 4);
    const char* wrow[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j)
        wrow[j] = reinterpret_cast<const char*>(a.g.b) + (long long)(8 * (j * 4 + wid) + r8) * a.g.ldb * 4 + bslot;
    auto issue_b = [&](int stage, int chunk) __attribute__((always_inline)) {
        const bool live = chunk < nchunks;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const char* src = live ? wrow[j] + (long long)chunk * 128 : reinterpret_cast<const char*>(a.g.zeros);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bring + stage * BSTAGE + (j * 4 + wid) * 1024), 16, 0, 0);
        }
    };
    // B fragment offsets inside a stage (rows n = j 32 + lrow; swizzle of a 32-aligned row block = (lrow >> 1) & 7)
    const int bswz = (lrow >> 1) & 7;
    int b_off[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_off[pl][ks] = lrow * 128 + (((pl * 4 + ks * 2 + lk) ^ bswz) << 4);

    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const int b = tile / a.tiles_per_sample;
        const int l0 = (tile - b * a.tiles_per_sample) * TL;
        // ---- weights of the first chunks go out first: their latency hides behind the A staging
#pragma unroll
        for (int s = 0; s < NSB - 1; ++s) issue_b(s, s);
        // ---- A image: rows r = 0 .. TL + 2H - 1 <-> positions l0 - H + r; 8 channels (32 B fp32 -> 16 B hi + 16 B lo) per piece.
        // All of a thread's loads go out before the first conversion (addresses clamped into the sample, values masked
        // afterwards), so a tile pays one memory latency, not one per piece.
        {
            const int rows = TL + 2 * H;
            constexpr int PPR = C / 8;                      // pieces per row
            constexpr int NPT = (AROWS * PPR + 255) / 256;  // pieces per thread (upper bound)
            const float* xb = a.x + (long long)b * a.L * C;
            float4 v0[NPT], v1[NPT];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int p0 = tid + 256 * u;
                const int r = p0 / PPR, q = p0 - r * PPR;
                int l = l0 - H + r;
                l = l < 0 ? 0 : (l >= a.L ? a.L - 1 : l);
                const float4* src = reinterpret_cast<const float4*>(xb + (long long)l * C + q * 8);
                v0[u] = src[0];
                v1[u] = src[1];
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int p0 = tid + 256 * u;
                const int r = p0 / PPR, q = p0 - r * PPR;
                const int l = l0 - H + r;
                if (r < rows) {
                    const float s = (l >= 0 && l < a.L) ? a.slope : 0.f;      // rows outside the sample: zeros (x * 0, then max(0, 0))
                    const float m = (l >= 0 && l < a.L) ? 1.f : 0.f;
                    float4 w0 = v0[u], w1 = v1[u];
                    w0.x = fmaxf(w0.x * m, w0.x * s);
                    w0.y = fmaxf(w0.y * m, w0.y * s);
                    w0.z = fmaxf(w0.z * m, w0.z * s);
                    w0.w = fmaxf(w0.w * m, w0.w * s);
                    w1.x = fmaxf(w1.x * m, w1.x * s);
                    w1.y = fmaxf(w1.y * m, w1.y * s);
                    w1.z = fmaxf(w1.z * m, w1.z * s);
                    w1.w = fmaxf(w1.w * m, w1.w * s);
                    unsigned h0, h1, h2, h3, q0, q1, q2, q3;
                    split2(w0.x, w0.y, h0, q0);
                    split2(w0.z, w0.w, h1, q1);
                    split2(w1.x, w1.y, h2, q2);
                    split2(w1.z, w1.w, h3, q3);
                    const u32x4 hi = {h0, h1, h2, h3}, lo = {q0, q1, q2, q3};
                    // channels 8q .. 8q+7 of line cb = q / 4: hi slot (q & 3), lo slot 4 + (q & 3), swizzled by the row
                    const int cb = q >> 2, sl = q & 3, sw = (r >> 1) & 7;
                    char* line = smem + (r * CB + cb) * 128;
                    *reinterpret_cast<u32x4*>(line + ((sl ^ sw) << 4)) = hi;
                    *reinterpret_cast<u32x4*>(line + (((4 + sl) ^ sw) << 4)) = lo;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        int st = 0, st_fill = NSB - 1;
        int tap = 0, cb = 0;
        for (int c = 0; c < nchunks; ++c) {
            wait_vmcnt<(NSB - 2) * IPW>();
            __builtin_amdgcn_s_barrier();          // chunk c of the weights is in LDS (and, at c = 0, the whole A image)
            issue_b(st_fill, c + NSB - 1);
            const char* bst = bring + st * BSTAGE;
            bf16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wid * (32 * MI) + i * 32 + lrow + tap * a.dil;      // image row of this lane's output position + tap
                const char* line = smem + (row * CB + cb) * 128;
                const int sw = (row >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    ah[ks][i] = *reinterpret_cast<const bf16x8*>(line + (((ks * 2 + lk) ^ sw) << 4));
                    al[ks][i] = *reinterpret_cast<const bf16x8*>(line + (((4 + ks * 2 + lk) ^ sw) << 4));
                }
            }
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bh[ks][j] = *reinterpret_cast<const bf16x8*>(bst + j * 4096 + b_off[0][ks]);
                    bl[ks][j] = *reinterpret_cast<const bf16x8*>(bst + j * 4096 + b_off[1][ks]);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
            }
            st = st + 1 == NSB ? 0 : st + 1;
            st_fill = st_fill + 1 == NSB ? 0 : st_fill + 1;
            if (++cb == CB) {
                cb = 0;
                ++tap;
            }
        }
        wait_vmcnt<0>();                           // (the trailing copies were dummies)
        // ---- epilogue: rows of this tile that exist (tail tile of a sample), all C columns
        IGemm q = a.g;
        const int mrow0 = b * a.L + l0;
        q.M = b * a.L + min(a.L, l0 + TL);
        igemm_epilogue<MI, NI>(q, acc, mrow0 + wid * (32 * MI), 0, lrow, lk, 0, C, 1);
        __builtin_amdgcn_s_barrier();              // everybody is done reading the image before the next tile overwrites it
    }
}

// ---------------------------------------------------------------------------------------------------------------- fused MRF pair
// One launch for the pair of an MRF resblock at the narrow stages (hifigan.py:54-61):
//     xt = c1(leaky(x));  xt = c2(leaky(xt));  x' = xt + x          (c1: k taps, dilation d1;  c2: k taps, dilation d2 = 1)
// The intermediate tensor never leaves the CU: 20 B of HBM traffic per element and pair (x in, t out, t in, x as residual, x' out)
// become 8 + the residual's re-read out of L2, and there is one staging pass and one launch instead of two.
//
// A tile is R1 = 128 MI rows of t -- positions l0 - H2 .. l0 - H2 + R1 - 1, H2 = d2 (k2 - 1) / 2 -- computed exactly as the
// one-convolution kernel computes them out of the staged image of leaky(x) (rows l0 - H2 - H1 ..).  The accumulators then
// take the first epilogue in registers (+ bias1, rows outside the sample -> 0: c2's zero padding), leaky-ReLU, hi / lo split,
// and are written over the x image as the split32 image of leaky(t) (the accumulator layout holds one column per lane: lane
// pairs swap halves so that every lane writes whole 4-byte words).  The second convolution reads it like the first read x;
// of its R1 output rows the first TLo = R1 - 2 H2 are positions l0 .. l0 + TLo - 1 and are stored through the shared
// epilogue (bias2, residual x, out_scale, accumulate); the last 2 H2 rows multiply rows past the image and are dropped.
// The weights of both convolutions stream through ONE ring as a single sequence of chunks.
// Arithmetic = the two launches': same products, same order, same fp32 epilogue in between -- bit-identical (tests).
struct HaloPairArgs {
    IGemm g;             // second convolution: b / ldb / bias, res, ldr, out_scale, accumulate, c, ldc, N, zeros
    const float* x;      // [B, L, C] fp32
    const float* w1;     // first convolution: packed split32 weights [C][k1 C], pitch ldb1 floats
    const float* bias1;
    int ldb1;
    int B, L, k1, d1, k2, d2;
    float slope1, slope2;
    int TLo, tiles_per_sample, tiles;
};

constexpr int HMAX2 = 16;          // rows the second convolution may reach past the t image (d2 (k2 - 1) <= HMAX2)

template <int C, int TL, int NSB>
__global__ __launch_bounds__(256) void halo_pair_kernel(const HaloPairArgs a) {
    constexpr int AROWS = TL + 2 * HMAX;       // x image (the t image, TL + HMAX2 rows, is written over it)
    constexpr int CB = C / 32;
    constexpr int NI = C / 32, MI = TL / 128;
    static_assert(TL == 128 || TL == 256, "tile length");
    static_assert(TL + HMAX2 <= AROWS, "t image inside the x image");
    constexpr int A_BYTES = AROWS * CB * 128;
    constexpr int BSTAGE = C * 128;
    constexpr int IPW = C / 32;
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [AROWS][CB][128] | [NSB][C][128]
    char* bring = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;
    const int H1 = a.d1 * (a.k1 - 1) / 2, H2 = a.d2 * (a.k2 - 1) / 2;
    const int n1 = a.k1 * CB, nchunks = n1 + a.k2 * CB;

    const int r8 = lane >> 3;
    const int bslot = (((lane & 7) ^ (((wid & 1) << 2) + (r8 >> 1))) << 4);
    const char *wrow1[IPW], *wrow2[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        wrow1[j] = reinterpret_cast<const char*>(a.w1) + (long long)(8 * (j * 4 + wid) + r8) * a.ldb1 * 4 + bslot;
        wrow2[j] = reinterpret_cast<const char*>(a.g.b) + (long long)(8 * (j * 4 + wid) + r8) * a.g.ldb * 4 + bslot;
    }
    auto issue_b = [&](int stage, int chunk) __attribute__((always_inline)) {
        const bool live = chunk < nchunks, first = chunk < n1;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const char* src = !live ? reinterpret_cast<const char*>(a.g.zeros)
                                    : first ? wrow1[j] + (long long)chunk * 128 : wrow2[j] + (long long)(chunk - n1) * 128;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bring + stage * BSTAGE + (j * 4 + wid) * 1024), 16, 0, 0);
        }
    };
    const int bswz = (lrow >> 1) & 7;
    int b_off[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_off[pl][ks] = lrow * 128 + (((pl * 4 + ks * 2 + lk) ^ bswz) << 4);

    for (int tile = blockIdx.x; tile < a.tiles; tile += gridDim.x) {
        const int b = tile / a.tiles_per_sample;
        const int l0 = (tile - b * a.tiles_per_sample) * a.TLo;
        const int t0 = l0 - H2;                    // position of row 0 of the t image
#pragma unroll
        for (int s = 0; s < NSB - 1; ++s) issue_b(s, s);
        // ---- x image: rows r <-> positions t0 - H1 + r, r < TL + 2 H1 (as the one-convolution kernel)
        {
            const int rows = TL + 2 * H1;
            constexpr int PPR = C / 8;
            constexpr int NPT = (AROWS * PPR + 255) / 256;
            const float* xb = a.x + (long long)b * a.L * C;
            float4 v0[NPT], v1[NPT];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int p0 = tid + 256 * u;
                const int r = p0 / PPR, q = p0 - r * PPR;
                int l = t0 - H1 + r;
                l = l < 0 ? 0 : (l >= a.L ? a.L - 1 : l);
                const float4* src = reinterpret_cast<const float4*>(xb + (long long)l * C + q * 8);
                v0[u] = src[0];
                v1[u] = src[1];
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int p0 = tid + 256 * u;
                const int r = p0 / PPR, q = p0 - r * PPR;
                const int l = t0 - H1 + r;
                if (r < rows) {
                    const float s = (l >= 0 && l < a.L) ? a.slope1 : 0.f;
                    const float m = (l >= 0 && l < a.L) ? 1.f : 0.f;
                    float4 w0 = v0[u], w1 = v1[u];
                    w0.x = fmaxf(w0.x * m, w0.x * s);
                    w0.y = fmaxf(w0.y * m, w0.y * s);
                    w0.z = fmaxf(w0.z * m, w0.z * s);
                    w0.w = fmaxf(w0.w * m, w0.w * s);
                    w1.x = fmaxf(w1.x * m, w1.x * s);
                    w1.y = fmaxf(w1.y * m, w1.y * s);
                    w1.z = fmaxf(w1.z * m, w1.z * s);
                    w1.w = fmaxf(w1.w * m, w1.w * s);
                    unsigned h0, h1, h2, h3, q0, q1, q2, q3;
                    split2(w0.x, w0.y, h0, q0);
                    split2(w0.z, w0.w, h1, q1);
                    split2(w1.x, w1.y, h2, q2);
                    split2(w1.z, w1.w, h3, q3);
                    const u32x4 hi = {h0, h1, h2, h3}, lo = {q0, q1, q2, q3};
                    const int cb = q >> 2, sl = q & 3, sw = (r >> 1) & 7;
                    char* line = smem + (r * CB + cb) * 128;
                    *reinterpret_cast<u32x4*>(line + ((sl ^ sw) << 4)) = hi;
                    *reinterpret_cast<u32x4*>(line + (((4 + sl) ^ sw) << 4)) = lo;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        f32x16 acc[MI][NI];
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        };
        zero_acc();

        int st = 0, st_fill = NSB - 1;
        int tap = 0, cb = 0, dil = a.d1;
        for (int c = 0; c < nchunks; ++c) {
            if (c == n1) {
                // ---- between the convolutions: t = acc + bias1 (the first launch's epilogue), rows outside the sample are c2's
                // zero padding, leaky-ReLU and split exactly as the second launch's staging did, written over the x image
                __builtin_amdgcn_s_barrier();      // every wave has read its last x rows
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int n = j * 32 + lrow;
                        const float bv = a.bias1 ? a.bias1[n] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wid * (32 * MI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                            const int l = t0 + row;
                            const bool in = l >= 0 && l < a.L;
                            float v = acc[i][j][r] * 1.0f + bv;
                            v = in ? fmaxf(v * 1.f, v * a.slope2) : 0.f;
                            const __bf16 hb16 = (__bf16)v;
                            const unsigned hb = __builtin_bit_cast(unsigned short, hb16);
                            const __bf16 lb16 = (__bf16)(v - __builtin_bit_cast(float, hb << 16));
                            const unsigned lb = __builtin_bit_cast(unsigned short, lb16);
                            // lanes 2m / 2m + 1 hold columns n / n + 1 of the same row: the even lane writes both hi halves, the odd
                            // lane both lo halves (one 4-byte LDS store per lane and element)
                            const bool even = (lane & 1) == 0;
                            const unsigned mine = even ? lb : hb;
                            const unsigned theirs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);
                            const unsigned word = even ? (hb | (theirs << 16)) : (theirs | (lb << 16));
                            // channel pair (n & ~1) of line j: 16-byte slot ((n & 31) >> 3) (+ 4 for the lo half), swizzled by the row
                            const int sl = ((n & 31) >> 3) + (even ? 0 : 4), sw = (row >> 1) & 7;
                            char* line = smem + (row * CB + j) * 128;
                            *reinterpret_cast<unsigned*>(line + ((sl ^ sw) << 4) + ((n & 6) << 1)) = word;
                        }
                    }
                // rows TL .. TL + 2 H2 - 1 of the t image feed only output rows that are dropped, but must be finite
                for (int e = tid; e < 2 * H2 * CB * 8; e += 256)
                    reinterpret_cast<u32x4*>(smem + TL * CB * 128)[e] = u32x4{0u, 0u, 0u, 0u};
                zero_acc();
                tap = 0;
                cb = 0;
                dil = a.d2;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            wait_vmcnt<(NSB - 2) * IPW>();
            __builtin_amdgcn_s_barrier();          // chunk c of the weights is in LDS (and, at c = 0 / n1, the whole image)
            issue_b(st_fill, c + NSB - 1);
            const char* bst = bring + st * BSTAGE;
            bf16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int row = wid * (32 * MI) + i * 32 + lrow + tap * dil;
                const char* line = smem + (row * CB + cb) * 128;
                const int sw = (row >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    ah[ks][i] = *reinterpret_cast<const bf16x8*>(line + (((ks * 2 + lk) ^ sw) << 4));
                    al[ks][i] = *reinterpret_cast<const bf16x8*>(line + (((4 + ks * 2 + lk) ^ sw) << 4));
                }
            }
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bh[ks][j] = *reinterpret_cast<const bf16x8*>(bst + j * 4096 + b_off[0][ks]);
                    bl[ks][j] = *reinterpret_cast<const bf16x8*>(bst + j * 4096 + b_off[1][ks]);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
            }
            st = st + 1 == NSB ? 0 : st + 1;
            st_fill = st_fill + 1 == NSB ? 0 : st_fill + 1;
            if (++cb == CB) {
                cb = 0;
                ++tap;
            }
        }
        wait_vmcnt<0>();
        // ---- second epilogue: output rows 0 .. TLo - 1 <-> positions l0 ..; rows of the next tile / past the sample are masked by M
        IGemm q = a.g;
        const int mrow0 = b * a.L + l0;
        q.M = b * a.L + min(a.L, l0 + a.TLo);
        igemm_epilogue<MI, NI>(q, acc, mrow0 + wid * (32 * MI), 0, lrow, lk, 0, C, 1);
        __builtin_amdgcn_s_barrier();
    }
}

template <int C, int TL, int NSB>
void launch_pair_c(const Ctx& ctx, const HaloPairArgs& a) {
    constexpr size_t lds = (size_t)(TL + 2 * HMAX) * (C / 32) * 128 + (size_t)NSB * C * 128;
    static_assert(lds <= 163840, "LDS per workgroup");
    auto kern = halo_pair_kernel<C, TL, NSB>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
    const int nc = device_cu_count(ctx.device);
    const int per_cu = (int)(163840 / lds) < 3 ? (int)(163840 / lds) : 3;
    long long grid = (long long)nc * (per_cu < 1 ? 1 : per_cu);
    if (grid > a.tiles) grid = a.tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx.stream, a);
}

template <int C, int TL, int NSB>
void launch_c(const Ctx& ctx, const HaloArgs& a) {
    constexpr size_t lds = (size_t)(TL + 2 * HMAX) * (C / 32) * 128 + (size_t)NSB * C * 128;
    static_assert(lds <= 163840, "LDS per workgroup");
    auto kern = halo_conv1d_kernel<C, TL, NSB>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
    const int nc = device_cu_count(ctx.device);
    const int per_cu = (int)(163840 / lds) < 3 ? (int)(163840 / lds) : 3;
    long long grid = (long long)nc * (per_cu < 1 ? 1 : per_cu);
    if (grid > a.tiles) grid = a.tiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx.stream, a);
}

}  // namespace

// Conv1d(C, C, k, dilation) with "same" padding on x [B, L, C] (fp32, channels-last), optional leaky-ReLU prologue, epilogue
// as conv_into (bias from the packed weight, residual, out_scale, accumulate).  false: shape / mode not covered -- the caller
// uses the generic implicit GEMM.  MAA_NO_HALO=1 disables it (A/B, bit-identity tests).
bool launch_halo_conv1d(const Ctx& ctx, const float* x, int B, int L, int C, const PackedW& w, int k, int dil, float slope,
                        const float* res, float out_scale, int accumulate, float* out) {
    const bool off = ctx.tune.halo == 0;
    // (C = 128 was tried with a 128-position tile, one workgroup per CU: 72.3 ms vs 70.6 ms for the generic engine on the
    //  config-3 stage -- at that width the layer is MFMA-bound and gains nothing from the staging; not kept)
    if (off || ctx.dtype != 1 || !(C == 32 || C == 64) || w.N != C || w.K != k * C || !w.split || !w.nk) return false;
    if (k < 1 || (k & 1) == 0 || dil < 1 || dil * (k - 1) / 2 > HMAX) return false;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (long long)B * L * C >= (1ll << 31)) return false;
    if (ctx.ws.dry) return true;
    HaloArgs a;
    a.g.b = w.w;
    a.g.ldb = w.ld;
    a.g.bias = w.bias;
    a.g.res = res;
    a.g.ldr = C;
    a.g.out_scale = out_scale;
    a.g.accumulate = accumulate;
    a.g.c = out;
    a.g.ldc = C;
    a.g.N = C;
    a.g.M = B * L;
    a.g.zeros = ctx.zeros;
    a.x = x;
    a.B = B;
    a.L = L;
    a.k = k;
    a.dil = dil;
    a.slope = slope;
    // tile length: 256 positions at C = 32 (56 KB of LDS, two workgroups per CU); 128 at C = 64 (80 KB: still two per CU, so
    // one workgroup's tile staging overlaps the other's MFMAs; 256 would be 112 KB and one per CU, and measured slower:
    // profiles/r2_halo_conv1d_variants.txt)
    const int TLr = C == 32 ? 256 : 128;
    a.tiles_per_sample = (L + TLr - 1) / TLr;
    a.tiles = B * a.tiles_per_sample;
    const double flops = 2.0 * B * (double)L * C * (double)C * k;
    ProfScope prof(ctx, C == 32 ? "halo_conv1d_bf16x3<32>" : "halo_conv1d_bf16x3<64>", flops, 12.0 * B * (double)L * C);
    if (C == 32)
        launch_c<32, 256, 4>(ctx, a);
    else
        launch_c<64, 128, 4>(ctx, a);
    MAA_HIP(hipGetLastError());
    return true;
}

// The MRF pair x' = (c2(leaky(c1(leaky(x)))) + res) * out_scale (+ out) in one launch; conditions as launch_halo_conv1d for both
// convolutions (same C, odd kernels, "same" padding), d2 (k2 - 1) <= HMAX2.  false: not covered -- the caller launches the two
// convolutions.  MAA_NO_PAIR=1 disables it (A/B, bit-identity tests).
bool launch_halo_pair(const Ctx& ctx, const float* x, int B, int L, int C, const PackedW& w1, int k1, int d1, float slope1,
                      const PackedW& w2, int k2, int d2, float slope2, const float* res, float out_scale, int accumulate,
                      float* out) {
    if (ctx.tune.halo != 2 || ctx.dtype != 1 || !(C == 32 || C == 64)) return false;
    for (const PackedW* w : {&w1, &w2})
        if (w->N != C || !w->split || !w->nk) return false;
    if (w1.K != k1 * C || w2.K != k2 * C) return false;
    if (k1 < 1 || (k1 & 1) == 0 || d1 < 1 || d1 * (k1 - 1) / 2 > HMAX) return false;
    if (k2 < 1 || (k2 & 1) == 0 || d2 < 1 || d2 * (k2 - 1) > HMAX2) return false;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0 || (long long)B * L * C >= (1ll << 31)) return false;
    if (ctx.ws.dry) return true;
    HaloPairArgs a;
    a.g.b = w2.w;
    a.g.ldb = w2.ld;
    a.g.bias = w2.bias;
    a.g.res = res;
    a.g.ldr = C;
    a.g.out_scale = out_scale;
    a.g.accumulate = accumulate;
    a.g.c = out;
    a.g.ldc = C;
    a.g.N = C;
    a.g.M = B * L;
    a.g.zeros = ctx.zeros;
    a.x = x;
    a.w1 = w1.w;
    a.bias1 = w1.bias;
    a.ldb1 = w1.ld;
    a.B = B;
    a.L = L;
    a.k1 = k1;
    a.d1 = d1;
    a.k2 = k2;
    a.d2 = d2;
    a.slope1 = slope1;
    a.slope2 = slope2;
    const int TLr = C == 32 ? 256 : 128;
    a.TLo = TLr - d2 * (k2 - 1);
    a.tiles_per_sample = (L + a.TLo - 1) / a.TLo;
    a.tiles = B * a.tiles_per_sample;
    const double flops = 2.0 * B * (double)L * C * (double)C * (k1 + k2);
    ProfScope prof(ctx, C == 32 ? "halo_pair_bf16x3<32>" : "halo_pair_bf16x3<64>", flops, 12.0 * B * (double)L * C);
    if (C == 32)
        launch_pair_c<32, 256, 4>(ctx, a);
    else
        launch_pair_c<64, 128, 4>(ctx, a);
    MAA_HIP(hipGetLastError());
    return true;
}

}  // namespace maa
