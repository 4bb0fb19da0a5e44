// Experimental lean variant of the 64x64 LDS-DMA kernel, kept in its own translation unit so that the measured
// kernels of igemm_dma.hip compile exactly as they were validated.  See the comment on the kernel.
#include "igemm_epilogue.h"

#include <cstdlib>

namespace maa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int NT = 256;
constexpr int BK = 32;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------------------------
// "Lean" 64x64 variant (EXPERIMENTAL, MAA_DMA_LEAN=1, default off: written after the GPU budget of round 1 was spent,
// to be measured first thing in round 2).  Same tiles, stages, barriers and arithmetic as igemm_dma_kernel<64,64,..>;
// what changes is the scalar/vector overhead around the 6 MFMAs of a chunk, which in that kernel is ~40 VALU + ~25
// SALU instructions per wave (about as many VALU-pipe cycles as the MFMAs take on the matrix pipe):
//   * every copy keeps a per-lane running pointer: one 64-bit add per copy and chunk instead of select + shift + add
//     + select (masked rows point at the zero page with a step of 0; pointers are rebuilt only when the tap changes);
//   * the wave index is made scalar (readfirstlane), so the LDS destinations (M0) are pure SALU;
//   * the K loop is unrolled over the NS stages, so stage offsets are ds_read / M0 immediates and the eight fragment
//     addresses are loop invariants.
template <int NS>
__global__ __launch_bounds__(NT) void igemm_dma_lean_kernel(const IGemm p, int ntiles, int Nb) {
    constexpr int BM = 64, BN = 64, WGN = 2, WTM = 32, WTN = 32;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;
    constexpr int IPW = ROWS / 32;             // 4
    static_assert(NS >= 2 && (NS - 2) * IPW <= 63, "stages");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    int nt, mt;
    if (p.m_fastest) {
        const int mtiles = gridDim.x / ntiles;
        mt = bid % mtiles;
        nt = bid / mtiles;
    } else {
        nt = bid % ntiles;
        mt = bid / ntiles;
    }
    const int m0 = mt * BM, n0 = nt * BN;

    const int Ctot = p.C1;
    const int rpb = p.Hout * p.Wout;
    const int Hlim = p.Hin << p.up, Wlim = p.Win << p.up;
    const int taps = p.KH * p.KW;
    const char* zero = reinterpret_cast<const char*>(p.zeros);
    const bool wave_is_a = wid * IPW * 8 < BM;  // waves 0,1 copy A rows, waves 2,3 copy B rows (uniform)

    // ---- this lane's IPW rows: running source pointer + per-chunk step (0 for masked rows)
    int a_b[IPW], a_iy0[IPW], a_ix0[IPW];
    const char* a_slot[IPW];
    const char* ptr[IPW];
    unsigned stp[IPW];                         // (unsigned: the 64-bit pointer add needs no sign extension)
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int row = 8 * (wid * IPW + j) + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        a_b[j] = -1;
        a_iy0[j] = a_ix0[j] = 0;
        a_slot[j] = reinterpret_cast<const char*>(p.a1) + slot * 16;
        ptr[j] = zero;
        stp[j] = 0;
        if (wave_is_a) {
            const int m = m0 + row;
            if (m < p.M) {
                const int b = m / rpb;
                const int rem = m - b * rpb;
                const int oy = rem / p.Wout;
                a_b[j] = b;
                a_iy0[j] = oy * p.sh - p.ph;
                a_ix0[j] = (rem - oy * p.Wout) * p.sw - p.pw;
            }
        } else {
            const int n = n0 + row - BM;
            if (n < Nb) {
                ptr[j] = reinterpret_cast<const char*>(p.b) + (long long)n * p.ldb * 4 + slot * 16;
                stp[j] = 128;
            }
        }
    }
    auto set_tap = [&](int tap) {              // A rows: pointer to channel 0 of the row under this tap
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            int iy = a_iy0[j] + ky * p.dh, ix = a_ix0[j] + kx * p.dw;
            const bool v = a_b[j] >= 0 && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim;
            iy >>= p.up;
            ix >>= p.up;
            const long long off = v ? ((long long)a_b[j] * p.Hin + iy) * p.Win + ix : 0;
            ptr[j] = v ? a_slot[j] + off * p.lda1 * 4 : zero;
            stp[j] = v ? 128u : 0u;
        }
    };
    int g_tap = 0, g_ci = 0;
    auto issue = [&](char* sbase) {            // sbase: this wave's 4 KB slice of the stage (scalar)
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)ptr[j], (lptr_t)(sbase + j * 1024), 16, 0, 0);
            ptr[j] += stp[j];
        }
        g_ci += BK;
        if (g_ci >= Ctot) {                    // next tap (A waves), or the end of K (everybody: copy zeros from now on)
            g_ci = 0;
            ++g_tap;
            if (g_tap >= taps) {
#pragma unroll
                for (int j = 0; j < IPW; ++j) {
                    ptr[j] = zero;
                    stp[j] = 0;
                }
            } else if (wave_is_a) {
                set_tap(g_tap);
            }
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int wm = wid / WGN, wn = wid - wm * WGN;
    const int lrow = lane & 31, lk = lane >> 5;
    const int swz = (lrow >> 1) & 7;
    const int a_row = (wm * WTM + lrow) * 128, b_row = (BM + wn * WTN + lrow) * 128;
    // loop-invariant LDS byte offsets of the eight fragments inside a stage: [k-step][al, bh, ah, bl]
    int fo[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int hi = ((0 * 4 + ks * 2 + lk) ^ swz) << 4, lo = ((1 * 4 + ks * 2 + lk) ^ swz) << 4;
        fo[ks][0] = a_row + lo;
        fo[ks][1] = b_row + hi;
        fo[ks][2] = a_row + hi;
        fo[ks][3] = b_row + lo;
    }
    auto compute = [&](const char* base) {     // base = smem + compile-time stage offset
        bf16x8 f[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 4; ++q) f[ks][q] = *reinterpret_cast<const bf16x8*>(base + fo[ks][q]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][0], f[ks][1], acc, 0, 0, 0);     // lo . hi
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][2], f[ks][3], acc, 0, 0, 0);     // hi . lo
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][2], f[ks][1], acc, 0, 0, 0);     // hi . hi
        }
    };

    const int nchunks = (p.K + BK - 1) / BK;
    if (wave_is_a) set_tap(0);
    char* wbase = smem + wid * (IPW * 1024);   // scalar
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(wbase + s * STAGE);
    for (int c = 0; c < nchunks; c += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {         // chunk c + u lives in stage u (nchunks is walked in whole rounds of NS)
            wait_vmcnt<(NS - 2) * IPW>();
            __builtin_amdgcn_s_barrier();
            issue(wbase + ((u + NS - 1) % NS) * STAGE);
            if (c + u < nchunks) compute(smem + u * STAGE);
        }
    }
    wait_vmcnt<0>();

    f32x16 accv[1][1];
    accv[0][0] = acc;
    igemm_epilogue<1, 1>(p, accv, m0 + wm * WTM, n0 + wn * WTN, lrow, lk, 0, Nb, rpb);
}

template <int NS>
void launch_lean(const Ctx& ctx, const IGemm& p, int Nb) {
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int mtiles = (p.M + 63) / 64, ntiles = (ncols + 63) / 64;
    dim3 grid((unsigned)((long long)mtiles * ntiles));
    constexpr size_t lds = (size_t)NS * 128 * 128;
    auto kern = igemm_dma_lean_kernel<NS>;
    static bool attr_set = false;
    if (!attr_set) {
        MAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx.stream, p, ntiles, Nb);
}

}  // namespace

void launch_igemm_dma_lean(const Ctx& ctx, const IGemm& p, int Nb) { launch_lean<4>(ctx, p, Nb); }

}  // namespace maa
