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
    const bool off = ctx.tune.no_halo;
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

}  // namespace maa
