// Implicit-GEMM engine on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, gfx950), fp32 in / fp32 out.
//
// Same problem description (IGemm), gather and epilogue as igemm_f32.hip; what changes is how the fp32 operands
// reach the MFMA:
//   TERMS = 3  "bf16x3": every fp32 operand is split on the fly into hi = bf16(x) and lo = bf16(x - hi) while its
//              tile is written to LDS, and the product is formed as lo*hi + hi*lo + hi*hi with fp32 accumulation.
//              hi+lo carries 16 mantissa bits, so a product is good to ~2^-16 instead of bf16's 2^-8, at 3 MFMAs
//              per 16-deep k-step: 16/3 = 5.3x the fp32 MFMA rate (gfx950 has no TF32/xf32, and its fp32 MFMA runs
//              at 1/16 of the bf16 rate, so this is the fast path that still meets the fp32 parity gates).
//   TERMS = 1  plain bf16 operands (throughput mode; error reported, not gated at 1e-4).
// LDS tiles are row-major with k contiguous: [row][32 k] bf16 = 64-byte rows, no padding, one plane for hi and one
// for lo; the four 16-byte slots of a row are XOR-swizzled by (row >> 2) & 3, so a lane's MFMA operand (8 consecutive
// k of one row) is one conflict-free ds_read_b128.  B must be given k-contiguous ([N][K]: packed weights in this
// mode, and K of Q.K^T).
// This is the REGISTER-STAGED engine (global -> VGPR -> LDS): it takes fp32 operands (split on the fly while the tile
// is written to LDS) and split32 operands alike.  Problems whose operands are BOTH split32 go to the LDS-DMA engine
// (igemm_dma.hip) in the bf16x3 mode; the two engines issue the same products in the same order.
#include "igemm_epilogue.h"

#include <cstdlib>

namespace maa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NT = 256;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);      // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, h);
}
// (a, b) -> packed hi pair and packed lo pair
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pk_bf16(a, b);
    const float fa = __builtin_bit_cast(float, hi << 16);
    const float fb = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pk_bf16(a - fa, b - fb);
}

// A_SPLIT / B_SPLIT: the operand already sits in memory pre-split ("split32" format: each group of 32 consecutive k
// of a row occupies one 128-byte line = [32 bf16 hi | 32 bf16 lo], so a row has the byte pitch of its fp32 form):
// GroupNorm / LayerNorm outputs and packed weights in the bf16 modes.  A tile row of one K chunk is then exactly one
// cache line, fetched once, and is copied global -> LDS as 16-byte pieces with no conversion.  Otherwise the operand
// is fp32 and is split while its tile is written to LDS (once per tile, i.e. once per tap and per N-tile for a
// conv -- which is why producers pre-split).
__device__ __forceinline__ void st16(unsigned short* dst, const float4& v) {   // 16-byte LDS store, by members
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 t = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<f32x4*>(dst) = t;
}

// BK: K depth of one LDS stage (32).  PF: register prefetch depth in K chunks (even).  The tile loads of chunk c + PF
// are issued before the MFMAs of chunk c and are first needed PF chunk-times later, when they are copied to LDS.
// Most of them miss L2 on first touch (a conv's weights alone are 3-4x one XCD's L2, and every XCD streams all of
// them), so a load takes ~2000 cycles: with PF = 2 a 64x64 chunk could not go faster than ~1000 cycles -- five times
// its MFMA time.  vmcnt retires in order, so only more register sets (or more waves) put more bytes in flight.
template <int BM, int BN, int WGM, int WGN, int TERMS, bool A_SPLIT, bool B_SPLIT, int BK, int PF>
__global__ __launch_bounds__(NT) void igemm_bf16_kernel(const IGemm p, int ntiles, int Nb) {
    // LDS rows are 64 B (32 bf16) with NO padding; the 16-byte chunk c of row r is stored at chunk c ^ ((r >> 2) & 3).
    // That XOR swizzle makes both the ds_write_b128 of the staging pass (8-lane groups = two rows = 32 distinct
    // banks) and the ds_read_b128 of the MFMA operands (16-lane groups {0-3,12-15,20-27} -> 16 distinct 16-byte
    // slots) conflict-free, and saves the 25 % of LDS a padded pitch costs.
    static_assert(BK == 32, "swizzle is written for 4 chunks per row");
    constexpr int LDK = BK;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int PLANES = TERMS == 1 ? 1 : 2;
    constexpr int ROWS = BM + BN;
    constexpr int PLANE_ELEMS = ROWS * LDK;            // bf16 elements of one plane of one buffer
    // per-thread 16-byte loads per chunk: fp32 operand: rows/32 float4; split operand: (rows/64 row slots) x planes
    // both operand forms: 8 threads per row, 16 bytes each = the 128-byte line of one (row, K chunk); 32 rows per pass
    constexpr int RPP = NT / 8;
    constexpr int ARS = BM / RPP, BRS = BN / RPP;     // row slots (= 16-byte loads) per thread per chunk
    constexpr int AL = ARS, BL = BRS;
    static_assert(WGM * WGN == 4 && BM % 32 == 0 && BN % 32 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];   // [buf][plane][row][LDK]

    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own 4 MB L2), so consecutive TILES are
    // handed to the SAME XCD -- the N-tiles of one M-tile and the neighbouring M-tiles (which share A rows through
    // the conv halo) then hit in one L2 instead of being fetched over the fabric once per XCD.  Bijective for any
    // grid size; a pure speed choice, results do not depend on placement.
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    int nt, mt;
    if (p.m_fastest) {             // an XCD's contiguous range = all M-tiles of a few N-tiles: weights fetched once chip-wide
        const int mtiles = gridDim.x / ntiles;
        mt = bid % mtiles;
        nt = bid / mtiles;
    } else {
        nt = bid % ntiles;
        mt = bid / ntiles;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int z = blockIdx.y;
    const int zo = z / p.zin, zi = z - zo * p.zin;
    const float* a1 = p.a1 + zo * p.a_so + zi * p.a_si;
    const float* a2 = p.a2;
    const float* bp = p.b + zo * p.b_so + zi * p.b_si;
    const long long coff = zo * p.c_so + zi * p.c_si;

    const int Ctot = p.C1 + p.C2;
    const int rpb = p.Hout * p.Wout;
    const int Hlim = p.Hin << p.up, Wlim = p.Win << p.up;
    const int taps = p.KH * p.KW;
    const float4* g_zero4 = reinterpret_cast<const float4*>(p.zeros);
    const float slope = p.a_act == 1 ? p.a_slope : 1.0f;

    // thread -> (row slot, k offset) for each operand form
    const int a_r0 = tid >> 3, b_r0 = tid >> 3;
    constexpr int a_rstep = RPP, b_rstep = RPP;
    const int seg = tid & 7;                 // 16-byte piece of the line: fp32 -> k = 4 seg; split32 -> plane seg>>2, chunk seg&3
    const int a_k = seg * 4, b_k = seg * 4;  // offset in fp32 units (4 bytes) for both forms

    // ---- A rows of this thread
    int a_b[ARS], a_iy0[ARS], a_ix0[ARS];
#pragma unroll
    for (int j = 0; j < ARS; ++j) {
        const int row = a_r0 + a_rstep * j;
        const int m = m0 + row;
        if (row < BM && m < p.M) {
            const int b = m / rpb;
            const int rem = m - b * rpb;
            const int oy = rem / p.Wout;
            const int ox = rem - oy * p.Wout;
            a_b[j] = b;
            a_iy0[j] = oy * p.sh - p.ph;
            a_ix0[j] = ox * p.sw - p.pw;
        } else {
            a_b[j] = -1;
            a_iy0[j] = 0;
            a_ix0[j] = 0;
        }
    }
    const void* a_p1[ARS];
    const void* a_p2[ARS];
    bool a_ok[ARS];
    int g_tap = 0, g_ci = 0;
    auto set_tap = [&](int tap) {
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int j = 0; j < ARS; ++j) {
            int iy = a_iy0[j] + ky * p.dh, ix = a_ix0[j] + kx * p.dw;
            const bool ok = a_b[j] >= 0 && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim;
            iy >>= p.up;
            ix >>= p.up;
            const long long off = ok ? ((long long)a_b[j] * p.Hin + iy) * p.Win + ix : 0;
            a_ok[j] = ok;
            a_p1[j] = a1 + off * p.lda1 + a_k;
            a_p2[j] = a2 + off * p.lda2 + a_k - p.C1;
        }
    };
    auto load_a = [&](float4 (&ra)[AL]) {
        const bool first = g_ci < p.C1;
        const int cend = first ? p.C1 : Ctot;
        const bool kin = A_SPLIT ? g_ci < cend : g_ci + a_k < cend;      // split32: whole lines only (C % 32 == 0)
#pragma unroll
        for (int j = 0; j < ARS; ++j) {
            const float4* src = reinterpret_cast<const float4*>(
                static_cast<const float*>(first ? a_p1[j] : a_p2[j]) + g_ci);
            const float4 v = *((a_ok[j] && kin) ? src : g_zero4);   // (a local first: a direct struct copy into
            ra[j] = v;                                               //  the array defeats SROA -> scratch)
        }
        g_ci += BK;
        if (g_ci >= Ctot && g_tap + 1 < taps) {
            g_ci = 0;
            ++g_tap;
            set_tap(g_tap);
        }
    };

    // ---- B rows ([N][K], k contiguous)
    const float* b_ptr[BRS];
    bool b_ok[BRS];
#pragma unroll
    for (int j = 0; j < BRS; ++j) {
        const int row = b_r0 + b_rstep * j;
        const int n = n0 + row;
        b_ok[j] = n < Nb;
        b_ptr[j] = bp + (long long)(b_ok[j] ? n : 0) * p.ldb + b_k;
    }
    auto load_b = [&](float4 (&rb)[BL], int k0) {
        const bool kin = B_SPLIT ? k0 < p.K : k0 + b_k < p.K;
#pragma unroll
        for (int j = 0; j < BRS; ++j) {
            const float4* src = reinterpret_cast<const float4*>(b_ptr[j] + k0);
            const float4 v = *((b_ok[j] && kin) ? src : g_zero4);
            rb[j] = v;
        }
    };

    // fp32 float4 (4 k) -> hi / lo words at LDS element offset e
    auto put_f32 = [&](unsigned short* base, int e, float4 v) {
        uint2 hi, lo;
        if constexpr (TERMS == 1) {
            hi.x = pk_bf16(v.x, v.y);
            hi.y = pk_bf16(v.z, v.w);
            *reinterpret_cast<uint2*>(base + e) = hi;
        } else {
            split2(v.x, v.y, hi.x, lo.x);
            split2(v.z, v.w, hi.y, lo.y);
            *reinterpret_cast<uint2*>(base + e) = hi;
            *reinterpret_cast<uint2*>(base + PLANE_ELEMS + e) = lo;
        }
    };
    // LDS element offset of this thread's piece of row `row` for a split32 line: plane seg>>2, chunk (seg&3) swizzled
    auto split_dst = [&](int row) { return (seg >> 2) * PLANE_ELEMS + row * LDK + (((seg & 3) ^ ((row >> 2) & 3)) << 3); };
    auto f32_dst = [&](int row) { return row * LDK + (((seg >> 1) ^ ((row >> 2) & 3)) << 3) + (seg & 1) * 4; };
    auto store_tiles = [&](const float4 (&ra)[AL], const float4 (&rb)[BL], int buf) {
        unsigned short* base = smem + buf * PLANES * PLANE_ELEMS;
#pragma unroll
        for (int j = 0; j < ARS; ++j) {
            const int row = a_r0 + a_rstep * j;
            if constexpr (A_SPLIT) {
                if (PLANES == 2 || seg < 4) st16(base + split_dst(row), ra[j]);
            } else {
                float4 v = ra[j];
                v.x = fmaxf(v.x, v.x * slope);
                v.y = fmaxf(v.y, v.y * slope);
                v.z = fmaxf(v.z, v.z * slope);
                v.w = fmaxf(v.w, v.w * slope);
                put_f32(base, f32_dst(row), v);
            }
        }
#pragma unroll
        for (int j = 0; j < BRS; ++j) {
            const int row = BM + b_r0 + b_rstep * j;
            if constexpr (B_SPLIT) {
                if (PLANES == 2 || seg < 4) st16(base + split_dst(row), rb[j]);
            } else {
                put_f32(base, f32_dst(row), rb[j]);
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid - wm * WGN;
    const int lrow = lane & 31, lk = lane >> 5;
    const int a_row = (wm * WTM + lrow) * LDK, b_row = (BM + wn * WTN + lrow) * LDK;
    const int sw = (lrow >> 2) & 3;          // row swizzle of this lane's rows (tile offsets are multiples of 32)

    auto compute = [&](int buf) {
        const unsigned short* base = smem + buf * PLANES * PLANE_ELEMS;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = ((ks * 2 + lk) ^ sw) << 3;
            bf16x8 ah[MI], bh[NI], al[MI], bl[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(base + a_row + i * 32 * LDK + ch);
                if constexpr (TERMS == 3) al[i] = *reinterpret_cast<const bf16x8*>(base + PLANE_ELEMS + a_row + i * 32 * LDK + ch);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8*>(base + b_row + j * 32 * LDK + ch);
                if constexpr (TERMS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(base + PLANE_ELEMS + b_row + j * 32 * LDK + ch);
            }
            // term-major order: the three products of one accumulator are separated by the other accumulators'
            // MFMAs (same per-accumulator order lo.hi, hi.lo, hi.hi, so results do not depend on the tile shape)
            if constexpr (TERMS == 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: PF register sets in flight, two LDS stages, one barrier per chunk
    static_assert(PF % 2 == 0 && PF >= 2, "prefetch depth");
    const int nchunks = (p.K + BK - 1) / BK;
    float4 ra[PF][AL], rb[PF][BL];
    set_tap(0);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        load_a(ra[u]);
        load_b(rb[u], u * BK);
    }
    store_tiles(ra[0], rb[0], 0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            // chunk c + u sits in LDS stage u & 1; set u is free again
            load_a(ra[u]);
            load_b(rb[u], (c + u + PF) * BK);
            if (c + u < nchunks) compute(u & 1);
            store_tiles(ra[(u + 1) % PF], rb[(u + 1) % PF], (u + 1) & 1);
            __syncthreads();
        }
    }

    igemm_epilogue<MI, NI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lrow, lk, coff, Nb, rpb);
}

template <int BM, int BN, int WGM, int WGN, int TERMS, bool AS, bool BS, int BKT, int PF>
void launch_one(const Ctx& ctx, const IGemm& p, int Nb) {
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (ncols + BN - 1) / BN;
    dim3 grid((unsigned)((long long)mtiles * ntiles), (unsigned)p.Z);
    constexpr int planes = TERMS == 1 ? 1 : 2;
    constexpr size_t lds = (size_t)2 * planes * (BM + BN) * BKT * sizeof(unsigned short);
    auto kern = igemm_bf16_kernel<BM, BN, WGM, WGN, TERMS, AS, BS, BKT, PF>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx.stream, p, ntiles, Nb);
}

template <int TERMS, bool AS, bool BS, int BKT>
void launch_tile(const Ctx& ctx, const IGemm& p, int cfg, int Nb) {
    // (PF = 4 was tried on the 64-wide tiles: no gain on L2-warm micro-benchmarks and 10 % slower inside the UNet,
    //  where its extra 32 VGPRs cost a wave per SIMD)
    switch (cfg) {
        case 0: launch_one<128, 128, 2, 2, TERMS, AS, BS, BKT, 2>(ctx, p, Nb); break;
        case 1: launch_one<128, 64, 2, 2, TERMS, AS, BS, BKT, 2>(ctx, p, Nb); break;
        case 2: launch_one<64, 64, 2, 2, TERMS, AS, BS, BKT, 2>(ctx, p, Nb); break;
        case 4: launch_one<128, 32, 4, 1, TERMS, AS, BS, BKT, 2>(ctx, p, Nb); break;
        default: launch_one<256, 32, 4, 1, TERMS, AS, BS, BKT, 2>(ctx, p, Nb); break;
    }
}

template <int TERMS, int BKT>
void launch_split(const Ctx& ctx, const IGemm& p, int cfg, int Nb) {
    if (p.a_split && p.b_split)
        launch_tile<TERMS, true, true, BKT>(ctx, p, cfg, Nb);
    else if (p.b_split)
        launch_tile<TERMS, false, true, BKT>(ctx, p, cfg, Nb);
    else
        launch_tile<TERMS, false, false, BKT>(ctx, p, cfg, Nb);
}

template <int TERMS>
void launch_terms(const Ctx& ctx, const IGemm& p, int cfg, int Nb) {
    launch_split<TERMS, 32>(ctx, p, cfg, Nb);
}

}  // namespace

// Returns false when the problem cannot take this path (B not k-contiguous, channel counts that are not a
// multiple of 32 under a multi-tap gather, unaligned rows): the caller then uses the fp32 kernel.
bool launch_igemm_bf16(const Ctx& ctx, const IGemm& p, int terms) {
    if (!p.b_nk) {
        MAA_CHECK(!p.a_split && !p.b_split, "split operands need the k-contiguous bf16 engine");
        return false;
    }
    const int taps = p.KH * p.KW, Ctot = p.C1 + p.C2;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    constexpr int BK = 32;      // channel granularity of the multi-tap gather (64-deep stages take two taps' worth)
    bool fast;
    if (p.a_split) {
        // split32 lines: whole 32-channel groups, rows pitched like their fp32 form
        fast = p.C2 == 0 && Ctot % 32 == 0 && p.lda1 % 32 == 0 && al16(p.a1) && p.Z == 1 && p.a_act == 0;
    } else {
        fast = (taps == 1 ? (p.C2 == 0 ? (Ctot % 4 == 0 || p.lda1 >= (Ctot + 3) / 4 * 4)
                                       : (p.C1 % BK == 0 && Ctot % 4 == 0))
                          : (Ctot % BK == 0 && p.C1 % BK == 0)) &&
               p.lda1 % 4 == 0 && al16(p.a1) && p.a_so % 4 == 0 && p.a_si % 4 == 0;
        if (p.C2 > 0) fast = fast && p.lda2 % 4 == 0 && al16(p.a2);
    }
    if (p.b_split)
        fast = fast && p.ldb % 32 == 0 && al16(p.b) && p.K % 32 == 0 && p.ldb >= p.K && p.Z == 1;
    else
        fast = fast && p.ldb % 4 == 0 && al16(p.b) && p.b_so % 4 == 0 && p.b_si % 4 == 0 && p.ldb >= (p.K + 3) / 4 * 4;
    fast = fast && p.K == taps * Ctot && (p.a_act == 0 || p.a_act == 1);
    if (!fast) {
        MAA_CHECK(!p.a_split && !p.b_split, "split operand given to a problem the bf16 engine cannot take");
        return false;
    }
    MAA_CHECK(!(p.c_split || p.c2) || p.N % 32 == 0, "split32 outputs are whole 32-channel lines");
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int Nb = ncols;       // rows of B that exist
    int cfg;
    if (p.geglu) {
        if (ncols % 64 != 0) return false;
        cfg = 0;
    } else if (ncols <= 32) {
        // 256-row tiles; 128-row ones while those would leave most of the chip idle (the UNet's 320 -> 4 output convolution:
        // 49 workgroups of 90 chunks each).  No K split either way: the two tiles give bit-identical results.
        cfg = (long long)((p.M + 255) / 256) * p.Z < 128 ? 4 : 3;
    } else {
        cfg = choose_tile(p.M, ncols, p.Z, true, ctx.kept_full() ? 1 : 0);
    }
    const bool no_dma = ctx.tune.no_dma;      // tests: same arithmetic, register staging
    // both bf16 modes: split32 x split32 problems go to the LDS-DMA engines (TERMS = 1: the hi halves are the bf16 operands)
    const bool dma = p.a_split && p.b_split && cfg < 3 && !no_dma;
    const PPPlan planp = dma ? igemm_pp_plan(ctx, p) : PPPlan();
    if (planp.bn) {
        // halo-staged ping-pong engine for the 3x3 convolutions (igemm_pp.hip); slabs borrowed like the second engine's
        const size_t mk = ctx.ws.mark();
        const size_t nf = igemm_pp_workspace_floats(p, planp);
        float* part = nf ? ctx.ws.alloc_f(nf) : nullptr;
        if (!ctx.ws.dry) {
            char shapep[64];
            const char* namep = igemm_pp_name(planp, terms);
            if (ctx.prof && ctx.prof->detail) {
                std::snprintf(shapep, sizeof(shapep), "pp%d M%d N%d K%d S%d", planp.bn, p.M, ncols, p.K, planp.S);
                namep = shapep;
            }
            ProfScope profp(ctx, namep, 2.0 * p.M * (double)ncols * p.K, 4.0 * ((double)p.K * ncols + (double)p.M * p.N));
            launch_igemm_pp(ctx, p, Nb, planp, part);
            MAA_HIP(hipGetLastError());
        }
        ctx.ws.release(mk);
        return true;
    }
    const PPPlan planq = dma && !planp.bn ? igemm_pp1_plan(ctx, p) : PPPlan();
    if (planq.bn) {
        const size_t mk = ctx.ws.mark();
        const size_t nf = igemm_pp1_workspace_floats(p, planq);
        float* part = nf ? ctx.ws.alloc_f(nf) : nullptr;
        if (!ctx.ws.dry) {
            char shapeq[64];
            const char* nameq = igemm_pp1_name(planq, terms);
            if (ctx.prof && ctx.prof->detail) {
                std::snprintf(shapeq, sizeof(shapeq), "pq%d M%d N%d K%d S%d", planq.bn, p.M, ncols, p.K, planq.S);
                nameq = shapeq;
            }
            ProfScope profq(ctx, nameq, 2.0 * p.M * (double)ncols * p.K, 4.0 * ((double)p.K * ncols + (double)p.M * p.N));
            launch_igemm_pp1(ctx, p, Nb, planq, part);
            MAA_HIP(hipGetLastError());
        }
        ctx.ws.release(mk);
        return true;
    }
    const Dma2Plan plan2 = dma ? igemm_dma2_plan(ctx, p) : Dma2Plan();
    if (plan2.cfg >= 0) {
        // wide tiles + split-K; the slabs are borrowed from the arena for the duration of the two launches (stream order
        // protects them from later borrowers)
        const size_t mk = ctx.ws.mark();
        const size_t nf = igemm_dma2_workspace_floats(p, plan2);
        float* part = nf ? ctx.ws.alloc_f(nf) : nullptr;
        if (!ctx.ws.dry) {
            const double flops2 = 2.0 * p.M * (double)ncols * p.K;
            const double bytes2 = 4.0 * ((double)p.K * ncols + (double)p.M * p.N);
            char shape2[64];
            const char* name2 = igemm_dma2_name(plan2, terms);
            if (ctx.prof && ctx.prof->detail) {
                std::snprintf(shape2, sizeof(shape2), "b2 M%d N%d K%d t%d", p.M, ncols, p.K, taps);
                name2 = shape2;
            }
            ProfScope prof2(ctx, name2, flops2, bytes2);
            launch_igemm_dma2(ctx, p, Nb, plan2, part);
            MAA_HIP(hipGetLastError());
        }
        ctx.ws.release(mk);
        return true;
    }
    if (ctx.ws.dry) return true;
    if (dma) cfg = igemm_dma_tile(p, cfg);
    const double flops = 2.0 * p.M * (double)ncols * p.K * p.Z;
    const double bytes = 4.0 * ((double)p.K * ncols + (double)p.M * p.N * p.Z);
    static const char* kNamesD[3] = {"igemm_dma_bf16x3<128x128>", "igemm_dma_bf16x3<128x64>", "igemm_dma_bf16x3<64x64>"};
    static const char* kNamesD1[3] = {"igemm_dma_bf16<128x128>", "igemm_dma_bf16<128x64>", "igemm_dma_bf16<64x64>"};
    static const char* kNames3[5] = {"igemm_bf16x3<128x128>", "igemm_bf16x3<128x64>", "igemm_bf16x3<64x64>", "igemm_bf16x3<256x32>", "igemm_bf16x3<128x32>"};
    static const char* kNames1[5] = {"igemm_bf16<128x128>", "igemm_bf16<128x64>", "igemm_bf16<64x64>", "igemm_bf16<256x32>", "igemm_bf16<128x32>"};
    char shape_name[48];
    const char* pname = dma ? (terms == 3 ? kNamesD[cfg] : kNamesD1[cfg]) : terms == 3 ? kNames3[cfg] : kNames1[cfg];
    if (ctx.prof && ctx.prof->detail) {
        std::snprintf(shape_name, sizeof(shape_name), "b%c%d M%d N%d K%d t%d Z%d", dma ? 'd' : 'g', cfg, p.M, ncols, p.K, taps, p.Z);
        pname = shape_name;
    }
    ProfScope prof(ctx, pname, flops, bytes);
    IGemm q = p;      // (tile order: N-tiles fastest inside an XCD's range -- M-fastest measured +4 % step time and was retired)
    if (dma)
        launch_igemm_dma(ctx, q, cfg, Nb);
    else if (terms == 3)
        launch_terms<3>(ctx, q, cfg, Nb);
    else
        launch_terms<1>(ctx, q, cfg, Nb);
    MAA_HIP(hipGetLastError());
    return true;
}

}  // namespace maa
