// bf16x3 implicit GEMM whose tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, gfx950), for problems where
// BOTH operands already sit in memory in the split32 form (GroupNorm / LayerNorm outputs against packed weights: the
// UNet's 3x3 convolutions and most of its linears).
//
// Why a second engine.  Ablation of the register-staged kernel (igemm_bf16.hip; profiles/r1_bf16x3_conv_ablation.txt):
// per 64x64 block and 32-deep K chunk it spends ~350 cycles bringing 16 KB global -> VGPR, ~250 cycles pushing the
// same 16 KB VGPR -> LDS (ds_write_b128 tops out at ~79 B/clk/CU), ~265 cycles on LDS -> VGPR fragment reads + 6
// MFMAs, and they hardly overlap: 880 cycles per chunk for 192 cycles of MFMA, at 2 or at 4 blocks per CU alike.
// With LDS-DMA a tile never touches the register file on its way in: no staging VGPRs, no ds_write pass, and the
// copies of the next NS-1 chunks stay in flight across the barriers (counted s_waitcnt vmcnt, raw s_barrier).
//
// LDS image of one stage: [BM + BN rows][128 B]; a row is the operand's split32 line for this K chunk (32 bf16 hi |
// 32 bf16 lo) with its eight 16-byte slots XOR-permuted by (row >> 1) & 7.  One DMA instruction moves 8 rows x 128 B:
// lane i lands at base + 16 i (the destination is lane-linear by construction of the instruction), so the permutation
// is applied to the lane's SOURCE address (slot (i & 7) ^ ((row >> 1) & 7) of row i >> 3) and again on the
// ds_read_b128 of the MFMA operands; every 8 lanes still fetch one whole 128-byte line.  Reads are conflict-free:
// the 16 lanes of a ds_read_b128 group hold rows {0-3, 12-15, 20-27} (+4, +32 ...), whose (row & 1, (row >> 1) & 7)
// pairs are all distinct, i.e. 16 different 16-byte slots of the 256-byte bank row.
//
// The arithmetic (per accumulator: lo.hi, hi.lo, hi.hi per 16-deep k-step, k ascending) is the register kernel's, so
// both engines return bit-identical results; MAA_NO_DMA=1 routes everything through the register kernel (tests).
#include "igemm_epilogue.h"

#include <cstdio>
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

// TERMS = 3: bf16x3 (lo.hi, hi.lo, hi.hi); TERMS = 1: the context's plain-bf16 mode -- the operands are the hi halves of the
// same split32 lines (hi = bf16(x) is exactly the rounding that mode asks for), one MFMA per k-step, the lo halves are never read
template <int BM, int BN, int WGM, int WGN, int NS, int TERMS>
__global__ __launch_bounds__(NT) void igemm_dma_kernel(const IGemm p, int ntiles, int Nb) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE = ROWS * 128;          // bytes
    constexpr int IPW = ROWS / 32;             // DMA instructions (8 rows x 128 B each) per wave and chunk
    static_assert(WGM * WGN == 4 && BM % 32 == 0 && BN % 32 == 0 && NS >= 2, "tile");
    static_assert((NS - 2) * IPW <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [NS][ROWS][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    // XCD-aware tile order (see igemm_bf16.hip)
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

    const int Ctot = p.C1;                     // single split32 source (checked by the launcher)
    const int rpb = p.Hout * p.Wout;
    const int Hlim = p.Hin << p.up, Wlim = p.Win << p.up;
    const int taps = p.KH * p.KW;
    const char* zero = reinterpret_cast<const char*>(p.zeros);

    // ---- the IPW tile rows this lane copies: row = 8 (wid IPW + j) + lane / 8, source slot (lane & 7) ^ swizzle.
    // (Every wave issues its share: one wave alone cannot feed the texture-address unit -- a dedicated loader wave
    //  measured 15 % slower than this arrangement.)
    bool is_a[IPW];                            // wave-uniform
    int a_b[IPW], a_iy0[IPW], a_ix0[IPW];
    const char* a_slot[IPW];                   // a1 + this lane's slot offset (tap independent)
    const char* src[IPW];                      // A: row pointer under the current tap; B: weight row
    bool ok[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int row = 8 * (wid * IPW + j) + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        is_a[j] = 8 * (wid * IPW + j) < BM;
        a_b[j] = -1;
        a_iy0[j] = a_ix0[j] = 0;
        a_slot[j] = reinterpret_cast<const char*>(p.a1) + slot * 16;
        src[j] = zero;
        ok[j] = false;
        if (is_a[j]) {
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
            ok[j] = n < Nb;
            src[j] = reinterpret_cast<const char*>(p.b) + (long long)(ok[j] ? n : 0) * p.ldb * 4 + slot * 16;
        }
    }
    auto set_tap = [&](int tap) {
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int j = 0; j < IPW; ++j)
            if (is_a[j]) {
                int iy = a_iy0[j] + ky * p.dh, ix = a_ix0[j] + kx * p.dw;
                const bool v = a_b[j] >= 0 && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim;
                iy >>= p.up;
                ix >>= p.up;
                const long long off = v ? ((long long)a_b[j] * p.Hin + iy) * p.Win + ix : 0;
                ok[j] = v;
                src[j] = a_slot[j] + off * p.lda1 * 4;
            }
    };
    int g_tap = 0, g_ci = 0;
    bool past = false;                          // chunks beyond K: everything masked (keeps the vmcnt arithmetic uniform)
    auto issue = [&](int stage) {
        char* sbase = smem + stage * STAGE + wid * (IPW * 1024);
        const int k0 = g_tap * Ctot + g_ci;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int off = is_a[j] ? g_ci : k0;
            const bool live = ok[j] && !past && (is_a[j] || k0 < p.K);
            const char* g = live ? src[j] + (long long)off * 4 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(sbase + j * 1024), 16, 0, 0);
        }
        g_ci += BK;
        if (g_ci >= Ctot) {
            g_ci = 0;
            ++g_tap;
            if (g_tap < taps)
                set_tap(g_tap);
            else
                past = true;
        }
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

    auto compute = [&](int stage) {
        const char* base = smem + stage * STAGE;
        // fragment reads first, then the MFMAs back to back (an instruction slotted between two MFMAs of one
        // accumulator chain costs ~40 cycles).  A wave with a single accumulator (64x64 tile) reads both k-steps up
        // front; with several accumulators per wave the reads of one k-step at a time keep the register count down.
        constexpr int G = MI * NI == 1 ? 2 : 1;          // k-steps per read group
#pragma unroll
        for (int k0 = 0; k0 < 2; k0 += G) {
            bf16x8 ah[G][MI], bh[G][NI], al[G][MI], bl[G][NI];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int ks = k0 + g;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    if constexpr (TERMS == 3) al[g][i] = *reinterpret_cast<const bf16x8*>(base + a_row + i * 4096 + slot_off[1][ks]);
#pragma unroll
                for (int j = 0; j < NI; ++j) bh[g][j] = *reinterpret_cast<const bf16x8*>(base + b_row + j * 4096 + slot_off[0][ks]);
#pragma unroll
                for (int i = 0; i < MI; ++i) ah[g][i] = *reinterpret_cast<const bf16x8*>(base + a_row + i * 4096 + slot_off[0][ks]);
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if constexpr (TERMS == 3) bl[g][j] = *reinterpret_cast<const bf16x8*>(base + b_row + j * 4096 + slot_off[1][ks]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if constexpr (TERMS == 3) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g][i], bh[g][j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g][i], bl[g][j], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g][i], bh[g][j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop.  Chunk c lives in stage c % NS.  Per chunk: wait until this wave's copies of chunk c have landed
    // (the NS-2 younger chunks may stay in flight), barrier (everybody's copies of chunk c are in LDS, and everybody is
    // done reading chunk c-1), refill the stage chunk c-1 used with chunk c+NS-1, then read + multiply chunk c.
    const int nchunks = (p.K + BK - 1) / BK;
    set_tap(0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    int st = 0, st_fill = NS - 1;
    for (int c = 0; c < nchunks; ++c) {
        wait_vmcnt<(NS - 2) * IPW>();
        __builtin_amdgcn_s_barrier();
        issue(st_fill);
        compute(st);
        st = st + 1 == NS ? 0 : st + 1;
        st_fill = st_fill + 1 == NS ? 0 : st_fill + 1;
    }
    wait_vmcnt<0>();        // no copy may land in LDS after this workgroup has given it back

    igemm_epilogue<MI, NI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lrow, lk, 0, Nb, rpb);
}

template <int BM, int BN, int WGM, int WGN, int NS, int TERMS>
void launch_terms(const Ctx& ctx, const IGemm& p, int Nb) {
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (ncols + BN - 1) / BN;
    dim3 grid((unsigned)((long long)mtiles * ntiles));
    constexpr size_t lds = (size_t)NS * (BM + BN) * 128;
    auto kern = igemm_dma_kernel<BM, BN, WGM, WGN, NS, TERMS>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx.stream, p, ntiles, Nb);
}

template <int BM, int BN, int WGM, int WGN, int NS>
void launch_one(const Ctx& ctx, const IGemm& p, int Nb) {
    if (ctx.dtype == 2)
        launch_terms<BM, BN, WGM, WGN, NS, 1>(ctx, p, Nb);
    else
        launch_terms<BM, BN, WGM, WGN, NS, 3>(ctx, p, Nb);
}

}  // namespace

// cfg: 0 = 128x128, 1 = 128x64, 2 = 64x64: the caller's tile choice passed through igemm_dma_tile().  The caller has
// checked the split32 conditions (both operands split, single source, C % 32 == 0, K % 32 == 0, 16-byte aligned rows,
// Z == 1, no A activation).
// Measured on the UNet's shapes (profiles/r1_bf16x3_dma_sweep.txt): 128x64 beats 64x64 by ~12 % once it still yields
// about two workgroups per CU and K is long (the 10x78-resolution convolutions); two LDS stages (more workgroups per
// CU) beat deeper copy queues except when there is only about one workgroup per CU (the 3x20-resolution layers).
int igemm_dma_tile(const IGemm& p, int cfg) {
    const long long ncols = (long long)p.N * (p.geglu ? 2 : 1);
    if (cfg == 2 && p.K >= 1024 && ((p.M + 127) / 128) * ((ncols + 63) / 64) >= 448) cfg = 1;
    return cfg;
}

void launch_igemm_dma(const Ctx& ctx, const IGemm& p, int cfg, int Nb) {
    const long long ncols = (long long)p.N * (p.geglu ? 2 : 1);
    // LDS stages: inside the UNet the weights of every layer come cold from HBM (each layer's weights are 3-4x an
    // XCD's L2 and the whole model streams through once per DDIM step), so the deeper copy queue wins there even
    // though the L2-warm micro-benchmark prefers more workgroups per CU: 64x64 -> 4 stages (64 KB, 2 workgroups/CU),
    // 128x64 -> 3 (72 KB, 2/CU), 128x128 -> 2 (64 KB, 2/CU).  In-pipeline A/B (20 DDIM steps + decode, ms): stages
    // (128x128, 128x64, 64x64) = (2,3,4) 234.0 | (2,2,2) +3.7 % | (2,3,5) +0.9 % | (2,3,6) +18.6 % | (2,4,4) +7.5 % | (3,3,4) +3.8 %.
    // Round 2: the long-K contractions moved to igemm_dma2.hip; on what is left to the 64x64 tile (K = 320 / 640 linears,
    // ten or twenty chunks) two stages -- five workgroups per CU instead of two -- win: in-pipeline (2,3,2) -2.3 % vs (2,3,4)
    // (profiles/r2_dma2_inpipe_probes.txt).
    int ns = cfg == 2 ? 2 : cfg == 1 ? 3 : 2;
    // Round 4: a 64x64 launch that puts fewer than ~2.5 workgroups on a CU (the K = 640 linears of the 5 x 39 level: 490 tiles
    // of 20 chunks) has one 16 KB chunk in flight per workgroup and waits ~1 us for each -- latency-, not fill-bound; such
    // launches take a deeper queue (three stages): one batch in flight 842.4 / 842.2 / 844.6 ms with two stages everywhere
    // against 839.1 (four) / 838.8 (three) in the same call (profiles/r4/r4_rowchain_ab_v1_serial_loop.txt).
    if (cfg == 2) {
        const long long tiles = (long long)((p.M + 63) / 64) * ((ncols + 63) / 64);
        if (tiles * 2 < 5LL * device_cu_count(ctx.device)) ns = 3;
    }
    switch (cfg) {
        case 0:
            launch_one<128, 128, 2, 2, 2>(ctx, p, Nb);
            break;
        case 1:
            launch_one<128, 64, 2, 2, 3>(ctx, p, Nb);
            break;
        default:
            if (ns >= 4) launch_one<64, 64, 2, 2, 4>(ctx, p, Nb);
            else if (ns == 3) launch_one<64, 64, 2, 2, 3>(ctx, p, Nb);
            else launch_one<64, 64, 2, 2, 2>(ctx, p, Nb);
            break;
    }
}

}  // namespace maa
