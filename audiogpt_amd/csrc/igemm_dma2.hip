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
// PIPE = true additionally software-pipelines the fragment reads inside a wave (barrier in mid-chunk).
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

template <int BM, int BN, int WGM, int WGN, int NS, bool PIPE, int PF>
__global__ __launch_bounds__(64 * WGM * WGN) void igemm_dma2_kernel(const IGemm p, int ntiles, int tiles, int Nb,
                                                                     int cps, float* __restrict__ part) {
    constexpr int NW = WGM * WGN, NTH = 64 * NW;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;                  // bytes
    constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;  // copies (8 rows x 128 B) per wave and chunk: A rows, B rows
    constexpr int IPW = AI + BI;
    // L2 prefetch (PF > 0): every chunk, each lane touches one 128-byte line of the tile rows of the chunk PF ahead with a
    // 4-byte LDS-DMA into a scratch area (no VGPR is written, so nothing waits for it).  The copies of that chunk, issued
    // PF - NS + 1 iterations later, then hit L2 instead of paying the MALL / HBM latency on the critical path: the LDS
    // stages alone cannot hold the latency x bandwidth product (42 B/clk x ~2900 clk = 120 KB for 128x128 tiles).
    constexpr int PFI = PF > 0 ? (ROWS + 64 * NW - 1) / (64 * NW) : 0;      // prefetch instructions per wave and chunk
    constexpr int VMI = IPW + PFI;                                          // VM operations per wave and chunk
    constexpr int VM_TAIL = PFI;        // prefetches issued after the copies of the same chunk (younger than them)
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 2 == 0 && BM % 64 == 0, "copy assignment");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % 32 == 0, "tile");
    static_assert(NS >= (PIPE ? 3 : 2) && (NS - 2) * VMI + VM_TAIL <= 63, "stages / vmcnt field");
    static_assert(PF == 0 || PF >= NS, "prefetch distance must exceed the copy queue");
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [NS][ROWS][128] (+ [NW][PFI][256] prefetch scratch)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- workgroup -> (K slice, tile).  An XCD (workgroup b runs on XCD b % 8) gets a contiguous range of
    // [slice][m tile][n tile], n fastest: the tiles sharing A rows / weight rows of one slice meet in one L2, and a
    // weight slice is fetched by 8 / S XCDs instead of all eight.
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
    const int slice = bid / tiles;
    const int tile = bid - slice * tiles;
    const int mt = tile / ntiles, nt = tile - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int Ctot = p.C1;                     // single split32 source (checked by the launcher)
    const int cpt = Ctot / BK;                 // chunks per tap
    const int rpb = p.Hout * p.Wout;
    const int Hlim = p.Hin << p.up, Wlim = p.Win << p.up;
    const int nchunks = p.K / BK;
    const int c_begin = slice * cps;
    const int c_end = min(nchunks, c_begin + cps);
    const char* zero = reinterpret_cast<const char*>(p.zeros);

    // ---- copies.  Copy q of an operand covers its rows 8q .. 8q+7; this wave issues q = j NW + wid.  Lane i moves the
    // 16 bytes at slot (i & 7) ^ swizzle of row i >> 3; the swizzle (row >> 1) & 7 = 4 (q & 1) + (i >> 4) is the same
    // for all of a wave's copies because NW is even.
    const int r8 = lane >> 3;
    const int slot_b = (((lane & 7) ^ (((wid & 1) << 2) + (r8 >> 1))) << 4);    // byte offset of the source slot
    int a_b[AI], a_iy0[AI], a_ix0[AI];
    const char* gpa[AI];
    unsigned inca[AI];
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
        gpa[j] = zero;
        inca[j] = 0;
    }
    const char* a_base = reinterpret_cast<const char*>(p.a1) + slot_b;
    auto set_tap = [&](int tap, int ci) {
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
    // weight rows past the last one are clamped to it: their columns are computed on real data and never stored
    // (a column of the product depends on its own weight row only), which keeps the step of every B copy uniform
    const char* gpb[BI];
    unsigned incb = BK * 4;
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int n = min(n0 + 8 * (j * NW + wid) + r8, Nb - 1);
        gpb[j] = reinterpret_cast<const char*>(p.b) + ((long long)n * p.ldb + (long long)c_begin * BK) * 4 + slot_b;
    }
    // ---- prefetch lines: line 64 (q NW + wid) + lane of the list [A rows | B rows] of the tile
    int pf_b[PFI > 0 ? PFI : 1], pf_iy0[PFI > 0 ? PFI : 1], pf_ix0[PFI > 0 ? PFI : 1];
    const char* pfp[PFI > 0 ? PFI : 1];
    unsigned pfinc[PFI > 0 ? PFI : 1];
    int pf_c = c_begin + PF, pf_tap = 0, pf_ci = 0;
    auto pf_set_tap = [&]() {
        const int ky = pf_tap / p.KW, kx = pf_tap - ky * p.KW;
#pragma unroll
        for (int q = 0; q < PFI; ++q) {
            if (64 * (q * NW + wid) < BM) {            // (wave-uniform)
                int iy = pf_iy0[q] + ky * p.dh, ix = pf_ix0[q] + kx * p.dw;
                const bool v = pf_b[q] >= 0 && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim;
                iy >>= p.up;
                ix >>= p.up;
                const long long pos = ((long long)pf_b[q] * p.Hin + iy) * p.Win + ix;
                pfp[q] = v ? reinterpret_cast<const char*>(p.a1) + (pos * p.lda1 + pf_ci) * 4 : zero;
                pfinc[q] = v ? BK * 4 : 0;
            }
        }
    };
    if constexpr (PFI > 0) {
        pf_tap = pf_c / cpt;
        pf_ci = (pf_c - pf_tap * cpt) * BK;
#pragma unroll
        for (int q = 0; q < PFI; ++q) {
            const int li = 64 * (q * NW + wid) + lane;
            pf_b[q] = -1;
            pf_iy0[q] = pf_ix0[q] = 0;
            pfp[q] = zero;
            pfinc[q] = 0;
            if (li < BM) {
                const int m = m0 + li;
                if (m < p.M) {
                    const int b = m / rpb;
                    const int rem = m - b * rpb;
                    const int oy = rem / p.Wout;
                    pf_b[q] = b;
                    pf_iy0[q] = oy * p.sh - p.ph;
                    pf_ix0[q] = (rem - oy * p.Wout) * p.sw - p.pw;
                }
            } else if (li < ROWS && pf_c < c_end) {
                const int n = min(n0 + li - BM, Nb - 1);
                pfp[q] = reinterpret_cast<const char*>(p.b) + ((long long)n * p.ldb + (long long)pf_c * BK) * 4;
                pfinc[q] = BK * 4;
            }
        }
        if (pf_c < c_end) pf_set_tap();
    }
    auto prefetch = [&]() {
        if constexpr (PFI > 0) {
            static_for<0, PFI>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                __builtin_amdgcn_global_load_lds((gptr_t)pfp[q], (lptr_t)(smem + NS * STAGE + (wid * PFI + q) * 256), 4, 0, 0);
                pfp[q] += pfinc[q];
            });
            ++pf_c;
            pf_ci += BK;
            if (pf_c >= c_end) {
#pragma unroll
                for (int q = 0; q < PFI; ++q) {
                    pfp[q] = zero;
                    pfinc[q] = 0;
                }
            } else if (pf_ci >= Ctot) {
                pf_ci = 0;
                ++pf_tap;
                pf_set_tap();
            }
        }
    };
    int g_tap = c_begin / cpt, g_ci = (c_begin - g_tap * cpt) * BK, g_c = c_begin;
    auto kill = [&]() {           // chunks past the slice: copies still issue (uniform vmcnt arithmetic), from the zero page
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            gpa[j] = zero;
            inca[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) gpb[j] = zero;
        incb = 0;
    };
    auto advance = [&]() {
        ++g_c;
        g_ci += BK;
        if (g_c >= c_end) {
            kill();
        } else if (g_ci >= Ctot) {
            g_ci = 0;
            ++g_tap;
            set_tap(g_tap, 0);
        }
    };
    // copy j of this wave's share of a chunk into stage `st` (j < AI: A rows, else B rows)
    auto copy1 = [&](int st, auto jc) {
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
    auto issue = [&](int st) {
        static_for<0, IPW>([&](auto jc) { copy1(st, jc); });
        advance();
        prefetch();
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
    auto read_frags = [&](const char* base, int ks, Frags& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i) f.al[i] = *reinterpret_cast<const bf16x8*>(base + a_row + i * 4096 + slot_off[1][ks]);
#pragma unroll
        for (int j = 0; j < NI; ++j) f.bh[j] = *reinterpret_cast<const bf16x8*>(base + b_row + j * 4096 + slot_off[0][ks]);
#pragma unroll
        for (int i = 0; i < MI; ++i) f.ah[i] = *reinterpret_cast<const bf16x8*>(base + a_row + i * 4096 + slot_off[0][ks]);
#pragma unroll
        for (int j = 0; j < NI; ++j) f.bl[j] = *reinterpret_cast<const bf16x8*>(base + b_row + j * 4096 + slot_off[1][ks]);
    };
    // the 3 MI NI MFMAs of one k-step; term-major, so the three products of an accumulator are MI NI - 1 MFMAs apart
    auto mma = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
    };

    if (g_c < c_end)
        set_tap(g_tap, g_ci);
    else
        kill();
    const int nloc = c_end - c_begin;          // chunks of this slice (>= 1 by construction of the grid)

    if constexpr (!PIPE) {
        // Chunk c lives in stage c % NS.  Per chunk: wait until this wave's copies of chunk c have landed (the NS-2
        // younger chunks stay in flight), barrier (everybody's copies of chunk c are in LDS and everybody is done
        // reading chunk c-1), refill the stage chunk c-1 used with chunk c+NS-1, then read + multiply chunk c.
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue(s);
        int st = 0, st_fill = NS - 1;
        for (int c = 0; c < nloc; ++c) {
            wait_vmcnt<(NS - 2) * VMI + VM_TAIL>();
            __builtin_amdgcn_s_barrier();
            if (!(p.dbg & 2)) issue(st_fill);
            if (!(p.dbg & 1)) {
                const char* base = smem + st * STAGE;
                Frags f0, f1;
                read_frags(base, 0, f0);
                read_frags(base, 1, f1);
                mma(f0);
                mma(f1);
            }
            st = st + 1 == NS ? 0 : st + 1;
            st_fill = st_fill + 1 == NS ? 0 : st_fill + 1;
        }
    } else {
        // Software-pipelined: the barrier sits in the middle of a chunk's MFMAs, so the fragment reads of the next
        // k-step, the barrier skew and the copy issue all hide under MFMAs of the same wave.
        //   P1(c): reads of (c, k-step 1) go out, then the MFMAs of (c, k-step 0) whose fragments were read in P2(c-1);
        //          wait for this wave's copies of chunk c+1; barrier B(c+1)
        //   P2(c): reads of (c+1, k-step 0) go out, then the MFMAs of (c, k-step 1), each followed by one of this
        //          wave's copies of chunk c+NS-1 into the stage chunk c-1 used (every wave consumed chunk c-1 -- its
        //          last reads fed the MFMAs of P2(c-1) -- before it reached B(c+1)).
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue(s);
        wait_vmcnt<(NS - 2) * VMI + VM_TAIL>();
        __builtin_amdgcn_s_barrier();              // chunk 0 is in LDS
        Frags f0, f1;
        read_frags(smem, 0, f0);
        int st = 0;
        for (int c = 0; c < nloc; ++c) {
            const char* cur = smem + st * STAGE;
            const int st_next = st + 1 == NS ? 0 : st + 1;
            const int st_fill = st == 0 ? NS - 1 : st - 1;
            read_frags(cur, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mma(f0);
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<(NS - 3) * VMI + VM_TAIL>();    // this wave's copies of chunk c+1 have landed
            __builtin_amdgcn_s_barrier();          // B(c+1): everybody's have, and everybody consumed chunk c-1
            read_frags(smem + st_next * STAGE, 0, f0);      // (past the last chunk: zeros, never multiplied)
            __builtin_amdgcn_sched_barrier(0);
            // MFMAs of k-step 1 interleaved with the IPW copies
            static_assert(3 * MI * NI >= IPW, "not enough MFMAs to carry the copies");
            static_for<0, 3 * MI * NI>([&](auto xc) {
                constexpr int x = decltype(xc)::value;
                constexpr int t = x / (MI * NI), i = (x % (MI * NI)) / NI, j = x % NI;
                if constexpr (t == 0)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.al[i], f1.bh[j], acc[i][j], 0, 0, 0);
                else if constexpr (t == 1)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.ah[i], f1.bl[j], acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1.ah[i], f1.bh[j], acc[i][j], 0, 0, 0);
                if constexpr (x < IPW) copy1(st_fill, xc);
                __builtin_amdgcn_sched_barrier(0);
            });
            advance();
            prefetch();
            st = st_next;
        }
    }
    wait_vmcnt<0>();        // no copy may land in LDS after this workgroup has given it back

    if (part == nullptr) {
        igemm_epilogue<MI, NI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lrow, lk, 0, Nb, rpb);
    } else {
        // slab of this (slice, tile): [MI NI blocks][4 register quads][NTH threads][4 floats] -- every store instruction
        // of a wave writes 1 KB contiguous
        float* pp = part + ((long long)(slice * tiles + tile) * (MI * NI * 4) * NTH + tid) * 4;
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
}

// Adds the S slabs of every 32x32 block in slice order (fixed: ((s0 + s1) + s2) + ...) and applies the epilogue.  A
// workgroup has the thread geometry of the GEMM workgroup and finishes block (i, j) -- with GEGLU the value / gate pair
// (j, j+1) -- of one tile, so every thread reads back exactly the registers its GEMM twin wrote.
template <int JW>
__global__ void splitk_reduce_kernel(const IGemm p, const float* __restrict__ part, int S, int tiles, int ntiles, int Nb,
                                     int BM, int BN, int WGN, int MI, int NI) {
    const int NTH = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int njb = NI / JW, nblk = MI * njb;
    const int tile = blockIdx.x / nblk, blk = blockIdx.x - tile * nblk;
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

template <int BM, int BN, int WGM, int WGN, int NS, bool PIPE, int PF>
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
    constexpr int PFI = PF > 0 ? (BM + BN + 64 * NW - 1) / (64 * NW) : 0;
    constexpr size_t lds = (size_t)NS * (BM + BN) * 128 + (size_t)NW * PFI * 256;
    static_assert(lds <= 163840, "LDS per workgroup");
    auto kern = igemm_dma2_kernel<BM, BN, WGM, WGN, NS, PIPE, PF>;
    static bool attr_set = false;
    if (!attr_set) {
        MAA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((long long)tiles * S)), dim3(NTH), lds, ctx.stream, p, ntiles, tiles, Nb, cps,
                       S > 1 ? part : nullptr);
    if (S > 1) {
        if constexpr (NI % 2 == 0) {
            if (p.geglu) {
                hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3((unsigned)(tiles * MI * (NI / 2))), dim3(NTH), 0, ctx.stream, p,
                                   part, S, tiles, ntiles, Nb, BM, BN, WGN, MI, NI);
                return;
            }
        }
        hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((unsigned)(tiles * MI * NI)), dim3(NTH), 0, ctx.stream, p, part, S, tiles,
                           ntiles, Nb, BM, BN, WGN, MI, NI);
    }
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

// Which problems take this engine, with which tile and how many K slices: a function of the layer (K, packed N) only.
// MAA_DMA2 = "off" | "cfg,ns,pipe,S[,kmin[,pf[,kmax]]]" overrides the policy for kmin <= K <= kmax (tuning and tests; read
// on every launch).
Dma2Plan plan_impl(const IGemm& p) {
    Dma2Plan pl;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int nchunks = p.K / BK;
    // MAA_DMA2_N<packed N> (e.g. MAA_DMA2_N320) overrides MAA_DMA2 for the layers of that width
    char var[32];
    std::snprintf(var, sizeof(var), "MAA_DMA2_N%d", ncols);
    const char* env = std::getenv(var);
    if (!env || !*env) env = std::getenv("MAA_DMA2");
    if (env && *env) {
        if (!std::strcmp(env, "off")) return pl;
        int cfg = 0, ns = 2, pipe = 0, S = 1, kmin = 0, pf = 0, kmax = 1 << 30;
        const int k = std::sscanf(env, "%d,%d,%d,%d,%d,%d,%d", &cfg, &ns, &pipe, &S, &kmin, &pf, &kmax);
        if (k >= 4) {
            if (p.K < kmin || (p.geglu && cfg >= 2)) return pl;
            if (p.K > kmax) env = nullptr;          // outside the override's K range: default policy below
            if (env) {
                pl.cfg = cfg;
                pl.ns = ns;
                pl.pipe = pipe;
                pl.pf = pf;
                pl.S = fit_slices(nchunks, S);
                return pl;
            }
        }
    }
    // default policy, from the sweeps of profiles/r2_dma2_sweep*.txt and the in-pipeline A/B of r2_dma2_inpipe.txt:
    //   * long-K contractions only (K >= 2048: the 3x3 convolutions and ff.net.2 at 5x39); below that the round-1 kernels
    //     win (ten or twenty chunks do not amortise the wide tile's prologue and the slab round trip);
    //   * N = 320 (the 10x78 level): one 128x320 tile holds all output channels -- no N padding (128x128 tiles waste 17 %
    //     there) and the weights are read once per M tile;
    //   * otherwise 128x128 tiles with the in-wave pipelined loop, four LDS stages;
    //   * two K slices: twice the workgroups for the 5x39 level's 125 tiles at the price of one slab round trip
    //     (more slices lose to the reduce traffic, fewer leave half the CUs idle).
    if (p.K < 2048 || ncols < 128 || p.geglu) return pl;
    if (ncols == 320) {
        pl.cfg = 2;
        pl.ns = 2;
        pl.pipe = 0;
    } else {
        pl.cfg = 0;
        pl.ns = 4;
        pl.pipe = 1;
    }
    pl.pf = 0;
    pl.S = fit_slices(nchunks, 2);
    return pl;
}

}  // namespace

Dma2Plan igemm_dma2_plan(const IGemm& p) { return plan_impl(p); }

size_t igemm_dma2_workspace_floats(const IGemm& p, const Dma2Plan& pl) {
    if (pl.cfg < 0 || pl.S <= 1) return 0;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int BM = pl.cfg == 1 ? 256 : pl.cfg == 3 ? 64 : 128, BN = pl.cfg >= 2 ? 320 : 128;
    const long long tiles = (long long)((p.M + BM - 1) / BM) * ((ncols + BN - 1) / BN);
    return (size_t)(tiles * pl.S * BM * BN);
}

const char* igemm_dma2_name(const Dma2Plan& pl) {
    static const char* names[4][2] = {{"igemm_dma2_bf16x3<128x128>", "igemm_dma2_bf16x3<128x128,splitK>"},
                                      {"igemm_dma2_bf16x3<256x128>", "igemm_dma2_bf16x3<256x128,splitK>"},
                                      {"igemm_dma2_bf16x3<128x320>", "igemm_dma2_bf16x3<128x320,splitK>"},
                                      {"igemm_dma2_bf16x3<64x320>", "igemm_dma2_bf16x3<64x320,splitK>"}};
    return names[pl.cfg < 0 || pl.cfg > 3 ? 0 : pl.cfg][pl.S > 1];
}

// The caller has checked the split32 conditions (both operands split, single source, C % 32 == 0, K % 32 == 0, 16-byte
// aligned rows, Z == 1, no A activation) and provides `part` = igemm_dma2_workspace_floats() floats when that is > 0.
void launch_igemm_dma2(const Ctx& ctx, const IGemm& p, int Nb, const Dma2Plan& pl, float* part) {
    MAA_CHECK(pl.cfg >= 0, "igemm_dma2: problem not planned for this engine");
    MAA_CHECK(pl.S == 1 || part != nullptr, "igemm_dma2: split-K needs its slab workspace");
    constexpr int D = 6;        // prefetch distance in chunks when the plan asks for L2 prefetch
    const int key = pl.cfg * 1000 + pl.ns * 100 + pl.pipe * 10 + (pl.pf ? 1 : 0);
    switch (key) {
        // 128x128 tiles, 4 waves of 64x64
        case 200: launch_one<128, 128, 2, 2, 2, false, 0>(ctx, p, Nb, pl.S, part); break;
        case 201: launch_one<128, 128, 2, 2, 2, false, D>(ctx, p, Nb, pl.S, part); break;
        case 300: launch_one<128, 128, 2, 2, 3, false, 0>(ctx, p, Nb, pl.S, part); break;
        case 301: launch_one<128, 128, 2, 2, 3, false, D>(ctx, p, Nb, pl.S, part); break;
        case 401: launch_one<128, 128, 2, 2, 4, false, D>(ctx, p, Nb, pl.S, part); break;
        case 310: launch_one<128, 128, 2, 2, 3, true, 0>(ctx, p, Nb, pl.S, part); break;
        case 410: launch_one<128, 128, 2, 2, 4, true, 0>(ctx, p, Nb, pl.S, part); break;
        case 411: launch_one<128, 128, 2, 2, 4, true, D>(ctx, p, Nb, pl.S, part); break;
        case 510: launch_one<128, 128, 2, 2, 5, true, 0>(ctx, p, Nb, pl.S, part); break;
        // 256x128 tiles, 8 waves of 64x64
        case 1200: launch_one<256, 128, 4, 2, 2, false, 0>(ctx, p, Nb, pl.S, part); break;
        case 1201: launch_one<256, 128, 4, 2, 2, false, D>(ctx, p, Nb, pl.S, part); break;
        case 1300: launch_one<256, 128, 4, 2, 3, false, 0>(ctx, p, Nb, pl.S, part); break;
        case 1310: launch_one<256, 128, 4, 2, 3, true, 0>(ctx, p, Nb, pl.S, part); break;
        case 1311: launch_one<256, 128, 4, 2, 3, true, D>(ctx, p, Nb, pl.S, part); break;
        // 128x320 tiles (all of N = 320 in one tile), 8 waves of 32x160
        case 2200: launch_one<128, 320, 4, 2, 2, false, 0>(ctx, p, Nb, pl.S, part); break;
        case 2201: launch_one<128, 320, 4, 2, 2, false, D>(ctx, p, Nb, pl.S, part); break;
        // 64x320 tiles, 4 waves of 32x160: short-K problems of the N = 320 / 640 layers (A read once, more bytes in flight)
        case 3200: launch_one<64, 320, 2, 2, 2, false, 0>(ctx, p, Nb, pl.S, part); break;
        case 3300: launch_one<64, 320, 2, 2, 3, false, 0>(ctx, p, Nb, pl.S, part); break;
        default: MAA_CHECK(false, "igemm_dma2: no such (tile, stages, pipe, prefetch) instantiation");
    }
}

}  // namespace maa
