// bf16x3 implicit GEMM, second LDS-DMA engine: wide tiles + split-K, for the UNet's long-K contractions.
//
// Why.  A tile row of one 32-deep K chunk is a 128-byte split32 line whatever the tile shape, so a BM x BN tile pulls
// 128 (BM + BN) bytes through the CU's texture-address path per chunk while its MFMAs keep the four SIMDs busy for
// BM BN 6 32 / (1024 4) cycles.  That path moves about 64 B/clk/CU (40 measured with the data in L2 / MALL):
//     64x64   16 KB per 192 cycles  = 85 B/clk   -> cannot be fed, whatever the schedule (round 1: 18 % of MFMA issue)
//     128x64  24 KB per 384 cycles  = 62 B/clk
//     128x128 32 KB per 768 cycles  = 42 B/clk
//     256x128 48 KB per 1536 cycles = 31 B/clk
// and the fragment traffic LDS -> VGPR drops from 1.33 KB to 0.67 KB per MFMA with 64x64 outputs per wave.  The UNet's
// problems are too small for such tiles to fill 256 CUs (M = 3120 x N = 640 is 125 tiles of 128x128), so K is cut into
// S slices that run as separate workgroups; each writes its fp32 accumulators as a slab, and splitk_reduce_kernel adds
// the slabs in slice order and applies the epilogue (bias, time-embedding row add, residual, GEGLU, split32 store).
//
// Numerical contract: S depends on the layer only (K and the packed N, never on M), so a sample's result does not
// depend on the batch it was computed in (tests: batch invariance, prompt sharding).  Within a slice the products are
// issued exactly as in igemm_bf16.hip / igemm_dma.hip (per accumulator lo.hi, hi.lo, hi.hi per 16-deep k-step, k
// ascending); with S = 1 the result is bit-identical to those engines.
//
// LDS image, swizzle and the DMA addressing are those of igemm_dma.hip (stage = [BM + BN rows][128 B], 16-byte slot s
// of row r stored at slot s ^ ((r >> 1) & 7), permutation applied to the copy's SOURCE address and to the ds_read_b128
// of the operands).  New here: every copy keeps a running per-lane source pointer (one 64-bit add per copy and chunk;
// masked rows point at the zero page with a step of 0; pointers are rebuilt only when the tap changes), the wave index
// is scalar so LDS destinations are SALU, and the A / B role of a copy is a compile-time property of its index.
// The main loop software-pipelines the fragment reads inside a wave (barrier in mid-chunk).  Since round 3 the UNet's 3x3
// convolutions run on igemm_pp.hip; this engine keeps ONE instantiation (128x128 tiles, four LDS stages) for what that one
// does not take: ff.net.2 at 5x39, strided / wide-image convolutions (the VAE's).  The other tiles it was swept over in
// round 2 (256x128, 128x320, 64x64, 128x64; two to five stages; unpipelined loop) are in profiles/r2_dma2_sweep*.txt.
// Workgroups are persistent: the grid is capped at what the chip holds at once and a workgroup runs its (slice, tile)
// items as one continuous stream of K chunks, so an item's cold start and store tail overlap its neighbours' MFMA work.
// (Tried and dropped, profiles/r2_dma2_sweep2*.txt: touching the lines of the chunk six ahead with 4-byte LDS-DMAs as an
//  L2 prefetch -- 0 to -3 % in the pipeline; a 64x320 tile for the short-K layers -- 14 % slower in the pipeline.)
#include "igemm_epilogue.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace maa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- indices into register arrays stay literal
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// TERMS: 3 = bf16x3; 1 = the context's plain-bf16 mode (operands = the hi halves of the same split32 lines, one MFMA per k-step)
template <int BM, int BN, int WGM, int WGN, int NS, int TERMS>
__global__ __launch_bounds__(64 * WGM * WGN) void igemm_dma2_kernel(const IGemm p, int ntiles, int tiles, int Nb,
                                                                     int cps, int items, float* __restrict__ part) {
    constexpr int NW = WGM * WGN, NTH = 64 * NW;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;                  // bytes
    constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;  // copies (8 rows x 128 B) per wave and chunk: A rows, B rows
    constexpr int IPW = AI + BI;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 2 == 0, "copy assignment");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % 32 == 0, "tile");
    static_assert(NS >= 3 && (NS - 2) * IPW <= 63, "stages / vmcnt field");
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [NS][ROWS][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- work items = (K slice, tile) pairs, `items` of them in the order [slice][m tile][n tile], n fastest.  An XCD
    // (workgroup b runs on XCD b % 8) owns a contiguous range of items -- the tiles sharing A rows / weight rows of one
    // slice meet in one L2, and a weight slice is fetched by 8 / S XCDs instead of all eight -- and its workgroups walk
    // that range with a stride of their number, so at any time they work on neighbouring items.
    // The grid may be smaller than `items` (persistent workgroups): a workgroup then runs its items as ONE stream of K
    // chunks -- the copies of the next item's first chunks are in flight while this item's epilogue runs, so the cold
    // start (address set-up, first-touch latency of the operands) and the store tail of an item overlap the neighbours'
    // MFMA work instead of adding up per round of workgroups.
    int w_lo, w_cnt, w_step;
    {
        const int G = gridDim.x, xcd = blockIdx.x & 7, qq = items >> 3, rr = items & 7;
        w_lo = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
        w_cnt = qq + (xcd < rr ? 1 : 0);
        w_step = (G - xcd + 7) >> 3;
    }
    const int w_first = (int)(blockIdx.x >> 3);
    if (w_first >= w_cnt) return;              // (never when grid <= items; uniform for the workgroup)

    const int Ctot = p.C1;                     // single split32 source (checked by the launcher)
    const int cpt = Ctot / BK;                 // chunks per tap
    const int rpb = p.Hout * p.Wout;
    const int Hlim = p.Hin << p.up, Wlim = p.Win << p.up;
    const int nchunks = p.K / BK;
    const char* zero = reinterpret_cast<const char*>(p.zeros);

    // ---- copies.  Copy q of an operand covers its rows 8q .. 8q+7; this wave issues q = j NW + wid.  Lane i moves the
    // 16 bytes at slot (i & 7) ^ swizzle of row i >> 3; the swizzle (row >> 1) & 7 = 4 (q & 1) + (i >> 4) is the same
    // for all of a wave's copies because NW is even.
    const int r8 = lane >> 3;
    const int slot_b = (((lane & 7) ^ (((wid & 1) << 2) + (r8 >> 1))) << 4);    // byte offset of the source slot
    int a_b[AI], a_iy0[AI], a_ix0[AI];
    const char* gpa[AI];
    unsigned inca[AI];
    const char* gpb[BI];
    unsigned incb = 0;
    const char* a_base = reinterpret_cast<const char*>(p.a1) + slot_b;
    int g_w = w_first, g_end = 0, g_tap = 0, g_ci = 0, g_c = 0;      // the item / chunk the next copies belong to
    auto set_tap = [&](int tap, int ci) __attribute__((always_inline)) {
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            int iy = a_iy0[j] + ky * p.dh, ix = a_ix0[j] + kx * p.dw;
            const bool v = a_b[j] >= 0 && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim;
            iy >>= p.up;
            ix >>= p.up;
            const long long pos = ((long long)a_b[j] * p.Hin + iy) * p.Win + ix;
            gpa[j] = v ? a_base + (pos * p.lda1 + ci) * 4 : zero;
            inca[j] = v ? BK * 4 : 0;
        }
    };
    // point the copies at the first chunk of item w
    auto setup_item = [&](int w) __attribute__((always_inline)) {
        const int item = w_lo + w;
        const int slice = item / tiles, tile = item - slice * tiles;
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        const int m0 = mt * BM, n0 = nt * BN;
        const int c_begin = slice * cps;
        g_end = min(nchunks, c_begin + cps);
        g_c = c_begin;
        g_tap = c_begin / cpt;
        g_ci = (c_begin - g_tap * cpt) * BK;
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            const int m = m0 + 8 * (j * NW + wid) + r8;
            a_b[j] = -1;
            a_iy0[j] = a_ix0[j] = 0;
            if (m < p.M) {
                const int b = m / rpb;
                const int rem = m - b * rpb;
                const int oy = rem / p.Wout;
                a_b[j] = b;
                a_iy0[j] = oy * p.sh - p.ph;
                a_ix0[j] = (rem - oy * p.Wout) * p.sw - p.pw;
            }
        }
        set_tap(g_tap, g_ci);
        // weight rows past the last one are clamped to it: their columns are computed on real data and never stored
        // (a column of the product depends on its own weight row only), which keeps the step of every B copy uniform
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int n = min(n0 + 8 * (j * NW + wid) + r8, Nb - 1);
            gpb[j] = reinterpret_cast<const char*>(p.b) + ((long long)n * p.ldb + (long long)c_begin * BK) * 4 + slot_b;
        }
        incb = BK * 4;
    };
    auto kill = [&]() __attribute__((always_inline)) {           // past the last item: copies still issue (uniform vmcnt arithmetic), from the zero page
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            gpa[j] = zero;
            inca[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) gpb[j] = zero;
        incb = 0;
    };
    auto advance = [&]() __attribute__((always_inline)) {
        ++g_c;
        g_ci += BK;
        if (g_c >= g_end) {
            g_w += w_step;
            if (g_w < w_cnt)
                setup_item(g_w);
            else
                kill();
        } else if (g_ci >= Ctot) {
            g_ci = 0;
            ++g_tap;
            set_tap(g_tap, 0);
        }
    };
    // copy j of this wave's share of a chunk into stage `st` (j < AI: A rows, else B rows)
    auto copy1 = [&](int st, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        char* sb = smem + st * STAGE + wid * 1024;
        if constexpr (j < AI) {
            __builtin_amdgcn_global_load_lds((gptr_t)gpa[j], (lptr_t)(sb + j * (NW * 1024)), 16, 0, 0);
            gpa[j] += inca[j];
        } else {
            constexpr int jb = j - AI;
            __builtin_amdgcn_global_load_lds((gptr_t)gpb[jb], (lptr_t)(sb + BM * 128 + jb * (NW * 1024)), 16, 0, 0);
            gpb[jb] += incb;
        }
    };
    auto issue = [&](int st) __attribute__((always_inline)) {
        static_for<0, IPW>([&](auto jc) { copy1(st, jc); });
        advance();
    };

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

    const int wm = wid / WGN, wn = wid - wm * WGN;
    const int lrow = lane & 31, lk = lane >> 5;
    const int swz = (lrow >> 1) & 7;           // tile offsets are multiples of 32 rows
    const int a_row = (wm * WTM + lrow) * 128, b_row = (BM + wn * WTN + lrow) * 128;
    int slot_off[2][2];                        // [plane][k-step] -> byte offset of this lane's 16-byte operand piece
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) slot_off[pl][ks] = ((pl * 4 + ks * 2 + lk) ^ swz) << 4;

    struct Frags {
        bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
    };
    auto read_frags = [&](const char* base, int ks, Frags& f) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if constexpr (TERMS == 3) f.al[i] = *reinterpret_cast<const bf16x8*>(base + a_row + i * 4096 + slot_off[1][ks]);
#pragma unroll
        for (int j = 0; j < NI; ++j) f.bh[j] = *reinterpret_cast<const bf16x8*>(base + b_row + j * 4096 + slot_off[0][ks]);
#pragma unroll
        for (int i = 0; i < MI; ++i) f.ah[i] = *reinterpret_cast<const bf16x8*>(base + a_row + i * 4096 + slot_off[0][ks]);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if constexpr (TERMS == 3) f.bl[j] = *reinterpret_cast<const bf16x8*>(base + b_row + j * 4096 + slot_off[1][ks]);
    };
    // the 3 MI NI MFMAs of one k-step; term-major, so the three products of an accumulator are MI NI - 1 MFMAs apart
    auto mma = [&](const Frags& f) __attribute__((always_inline)) {
        if constexpr (TERMS == 3) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
    };

    // ---- the item whose chunks are being multiplied
    int c_w = w_first, c_rem = 0;
    auto item_chunks = [&](int w) __attribute__((always_inline)) {
        const int slice = (w_lo + w) / tiles;
        return min(nchunks, (slice + 1) * cps) - slice * cps;
    };
    // epilogue (or slab store) of the finished item, then on to the next; false: this workgroup is done
    auto finish_item = [&]() __attribute__((always_inline)) {
        const int item = w_lo + c_w;
        const int slice = item / tiles, tile = item - slice * tiles;
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        if (part == nullptr) {
            igemm_epilogue<MI, NI>(p, acc, mt * BM + wm * WTM, nt * BN + wn * WTN, lrow, lk, 0, Nb, rpb);
        } else {
            // slab of this (slice, tile): [MI NI blocks][4 register quads][NTH threads][4 floats] -- every store
            // instruction of a wave writes 1 KB contiguous
            float* pp = part + ((long long)item * (MI * NI * 4) * NTH + tid) * 4;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(pp + (long long)((i * NI + j) * 4 + q) * NTH * 4) = v;
                    }
        }
        zero_acc();
        c_w += w_step;
        if (c_w >= w_cnt) return false;
        c_rem = item_chunks(c_w);
        return true;
    };

    setup_item(g_w);
    c_rem = item_chunks(c_w);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);

    {
        // Software-pipelined: the barrier sits in the middle of a chunk's MFMAs, so the fragment reads of the next
        // k-step, the barrier skew and the copy issue all hide under MFMAs of the same wave.
        //   P1(c): reads of (c, k-step 1) go out, then the MFMAs of (c, k-step 0) whose fragments were read in P2(c-1);
        //          wait for this wave's copies of chunk c+1; barrier B(c+1)
        //   P2(c): reads of (c+1, k-step 0) go out, then the MFMAs of (c, k-step 1), each followed by one of this
        //          wave's copies of chunk c+NS-1 into the stage chunk c-1 used (every wave consumed chunk c-1 -- its
        //          last reads fed the MFMAs of P2(c-1) -- before it reached B(c+1)).
        wait_vmcnt<(NS - 2) * IPW>();
        __builtin_amdgcn_s_barrier();              // chunk 0 is in LDS
        Frags f0, f1;
        read_frags(smem, 0, f0);
        int st = 0;
        for (;;) {
            const char* cur = smem + st * STAGE;
            const int st_next = st + 1 == NS ? 0 : st + 1;
            const int st_fill = st == 0 ? NS - 1 : st - 1;
            read_frags(cur, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mma(f0);
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<(NS - 3) * IPW>();          // this wave's copies of chunk c+1 have landed
            __builtin_amdgcn_s_barrier();          // B(c+1): everybody's have, and everybody consumed chunk c-1
            read_frags(smem + st_next * STAGE, 0, f0);      // (past the last chunk: zeros, never multiplied)
            __builtin_amdgcn_sched_barrier(0);
            // MFMAs of k-step 1 interleaved with the IPW copies
            static_for<0, TERMS * MI * NI>([&](auto xc) {
                constexpr int x = decltype(xc)::value;
                constexpr int t = x / (MI * NI) + (3 - TERMS), i = (x % (MI * NI)) / NI, j = x % NI;
                if constexpr (t == 0)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.al[i], f1.bh[j], acc[i][j], 0, 0, 0);
                else if constexpr (t == 1)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.ah[i], f1.bl[j], acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.ah[i], f1.bh[j], acc[i][j], 0, 0, 0);
                if constexpr (x < IPW) copy1(st_fill, xc);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (IPW > TERMS * MI * NI)          // (small tiles, one term: more copies than MFMAs to carry them)
                static_for<TERMS * MI * NI, IPW>([&](auto jc) { copy1(st_fill, jc); });
            advance();
            st = st_next;
            if (--c_rem == 0)
                if (!finish_item()) break;
        }
    }
    wait_vmcnt<0>();        // no copy may land in LDS after this workgroup has given it back
}

// Adds the S slabs of every 32x32 block in slice order (fixed: ((s0 + s1) + s2) + ...) and applies the epilogue.  A
// workgroup has the thread geometry of the GEMM workgroup and finishes block (i, j) -- with GEGLU the value / gate pair
// (j, j+1) -- of one tile, so every thread reads back exactly the registers its GEMM twin wrote.
template <int JW>
__global__ void splitk_reduce_kernel(const IGemm p, const float* __restrict__ part, int S, int tiles, int ntiles, int Nb,
                                     int BM, int BN, int WGN, int MI, int NI, int xcd_on) {
    const int NTH = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int njb = NI / JW, nblk = MI * njb;
    const int wi = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, xcd_on);      // the blocks of a tile, and neighbouring tiles' rows, on one XCD
    const int tile = wi / nblk, blk = wi - tile * nblk;
    const int i = blk / njb, j = (blk - i * njb) * JW;
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const int wm = wid / WGN, wn = wid - wm * WGN;
    const int WTM = 32 * MI, WTN = 32 * NI;
    f32x16 acc[1][JW];
    const long long slab = (long long)(MI * NI * 4) * NTH * 4;          // floats per (slice, tile)
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
        const float* src = part + (long long)tile * slab + ((long long)((i * NI + j + jj) * 4) * NTH + tid) * 4;
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(src + (long long)q * NTH * 4);
        for (int s = 1; s < S; ++s) {
            const float* sp = src + (long long)s * tiles * slab;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += *reinterpret_cast<const f32x4*>(sp + (long long)q * NTH * 4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][jj][4 * q + e] = v[q][e];
    }
    const int rpb = p.Hout * p.Wout;
    igemm_epilogue<1, JW>(p, acc, mt * BM + wm * WTM + i * 32, nt * BN + wn * WTN + j * 32, lane & 31, lane >> 5, 0, Nb, rpb);
}

// Persistent grids: at most this many workgroups per CU's worth of LDS, times the CUs.  MAA_DMA2_PERSIST=0 launches one
// workgroup per work item instead (A/B).
int cu_count(const Ctx& ctx) { return device_cu_count(ctx.device); }

template <int BM, int BN, int WGM, int WGN, int NS>
void launch_one(const Ctx& ctx, const IGemm& p, int Nb, int S, float* part) {
    constexpr int NW = WGM * WGN, NTH = 64 * NW;
    constexpr int MI = BM / WGM / 32, NI = BN / WGN / 32;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (ncols + BN - 1) / BN;
    const int tiles = mtiles * ntiles;
    const int nchunks = p.K / BK;
    const int cps = (nchunks + S - 1) / S;
    const int Seff = (nchunks + cps - 1) / cps;            // slices that have at least one chunk
    MAA_CHECK(Seff == S, "split-K plan leaves an empty slice");
    MAA_CHECK(!p.geglu || NI % 2 == 0, "GEGLU needs value / gate block pairs inside a wave");
    constexpr size_t lds = (size_t)NS * (BM + BN) * 128;
    static_assert(lds <= 163840, "LDS per workgroup");
    const long long items = (long long)tiles * S;
    long long grid = items;
    {
        const int per_cu = (int)(163840 / lds) < 32 / NW ? (int)(163840 / lds) : 32 / NW;      // LDS- and wave-limited residency
        const long long cap = (long long)cu_count(ctx) * (per_cu < 1 ? 1 : per_cu);
        if (grid > cap) grid = cap;
    }
    auto go = [&](auto kern) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTH), lds, ctx.stream, p, ntiles, tiles, Nb, cps, (int)items,
                           S > 1 ? part : nullptr);
    };
    if (ctx.dtype == 2)
        go(igemm_dma2_kernel<BM, BN, WGM, WGN, NS, 1>);
    else
        go(igemm_dma2_kernel<BM, BN, WGM, WGN, NS, 3>);
    if (S > 1) launch_splitk_reduce(ctx, p, part, S, tiles, ntiles, Nb, BM, BN, WGN, MI, NI, NTH);
}

// slices for a K of `nchunks` 32-deep chunks such that no slice is empty: the largest S' <= S with
// ceil(nchunks / ceil(nchunks / S')) == S'
int fit_slices(int nchunks, int S) {
    if (S < 1) S = 1;
    if (S > nchunks) S = nchunks;
    for (; S > 1; --S) {
        const int cps = (nchunks + S - 1) / S;
        if ((nchunks + cps - 1) / cps == S) break;
    }
    return S;
}

// Which problems take this engine and with how many K slices: a function of the layer (K, packed N) only.
// MAA_DMA2 = "off" | "0,4,1,S[,kmin[,kmax]]" (tile, stages, pipelining -- one instantiation is kept -- and K slices) overrides
// the policy for kmin <= K <= kmax (tests; parsed when the context is created).
Dma2Plan plan_impl(const Ctx& ctx, const IGemm& p) {
    Dma2Plan pl;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int nchunks = p.K / BK;
    const std::string* env = ctx.tune.dma2.empty() ? nullptr : &ctx.tune.dma2;
    if (env) {
        if (*env == "off") return pl;
        int cfg = 0, ns = 4, pipe = 1, S = 1, kmin = 0, kmax = 1 << 30;
        const int k = std::sscanf(env->c_str(), "%d,%d,%d,%d,%d,%d", &cfg, &ns, &pipe, &S, &kmin, &kmax);
        if (k >= 4) {      // (format validated by Tuning::load when the context was created)
            if (p.K < kmin) return pl;
            if (p.K <= kmax) {
                pl.cfg = 0;
                pl.S = fit_slices(nchunks, S);
                return pl;
            }
        }
    }
    // default policy (profiles/r2_dma2_sweep*.txt, r2_dma2_inpipe*.txt): long-K contractions only (K >= 2048:
    // what the ping-pong engine does not take -- ff.net.2 at 5x39, the VAE's wide 3x3 convolutions); 128x128 tiles, four LDS
    // stages, in-wave pipelined loop; two K slices (more lose to the reduce traffic, fewer leave half the CUs idle at 5x39).
    if (p.K < 2048 || ncols < 128 || p.geglu) return pl;
    pl.cfg = 0;
    pl.S = fit_slices(nchunks, 2);
    return pl;
}

}  // namespace

void launch_splitk_reduce(const Ctx& ctx, const IGemm& p, const float* part, int S, int tiles, int ntiles, int Nb, int BM,
                          int BN, int WGN, int MI, int NI, int NTH) {
    if (p.geglu && NI % 2 == 0) {
        hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3((unsigned)(tiles * MI * (NI / 2))), dim3(NTH), 0, ctx.stream, p, part, S,
                           tiles, ntiles, Nb, BM, BN, WGN, MI, NI, 1);
        return;
    }
    MAA_CHECK(!p.geglu, "split-K reduce: GEGLU needs value / gate block pairs inside a wave");
    hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((unsigned)(tiles * MI * NI)), dim3(NTH), 0, ctx.stream, p, part, S, tiles,
                       ntiles, Nb, BM, BN, WGN, MI, NI, 1);
}

Dma2Plan igemm_dma2_plan(const Ctx& ctx, const IGemm& p) { return plan_impl(ctx, p); }

size_t igemm_dma2_workspace_floats(const IGemm& p, const Dma2Plan& pl) {
    if (pl.cfg < 0 || pl.S <= 1) return 0;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const long long tiles = (long long)((p.M + 127) / 128) * ((ncols + 127) / 128);
    return (size_t)(tiles * pl.S * 128 * 128);
}

const char* igemm_dma2_name(const Dma2Plan& pl, int terms) {
    if (terms == 1) return pl.S > 1 ? "igemm_dma2_bf16<128x128,splitK>" : "igemm_dma2_bf16<128x128>";
    return pl.S > 1 ? "igemm_dma2_bf16x3<128x128,splitK>" : "igemm_dma2_bf16x3<128x128>";
}

// The caller has checked the split32 conditions (both operands split, single source, C % 32 == 0, K % 32 == 0, 16-byte
// aligned rows, Z == 1, no A activation) and provides `part` = igemm_dma2_workspace_floats() floats when that is > 0.
void launch_igemm_dma2(const Ctx& ctx, const IGemm& p, int Nb, const Dma2Plan& pl, float* part) {
    MAA_CHECK(pl.cfg == 0, "igemm_dma2: problem not planned for this engine");
    MAA_CHECK(pl.S == 1 || part != nullptr, "igemm_dma2: split-K needs its slab workspace");
    launch_one<128, 128, 2, 2, 4>(ctx, p, Nb, pl.S, part);      // 4 waves of 64x64, four LDS stages, in-wave pipelined loop
}

}  // namespace maa
