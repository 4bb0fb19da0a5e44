// Row-chain engine (round 4): the short-K linears of a SpatialTransformer at the 10 x 78 level, fused per 64-row block.
//
//   stage 1   y = A . W1^T + b1 (+ res1)                 A: split32 rows [M, K1], W1: packed split32 [N][K1]
//   between   t = LayerNorm(y) (gamma, beta, eps)        or t = y;  y optionally written as fp32, t optionally as split32
//   stage 2   z = t . W2^T + b2 (+ res2)                 W2: packed split32 [T2 N][N]; z fp32 or split32
//
// replaces, per BasicTransformerBlock (ldm/modules/attention.py:196-215, 250-261 in the reference):
//   proj_in -> norm1 -> to_q|to_k|to_v     (one launch instead of GEMM, LayerNorm, GEMM)
//   attn1.to_out (+x) -> norm2 -> attn2.to_q
//   attn2.to_out (+x) -> norm3             (the GEGLU projection keeps its own engine)
//   ff.net.2 (+x) -> proj_out (+x_in)
//
// Why.  These contractions are 2.6 GFLOP against 48 MB of operand + result bytes each: one 64 x 64-tile launch costs
// ~19 us of which ~10 are ramp-up and the fp32 epilogue, the LayerNorm between two of them is a full read + write of the
// row block, and every launch re-reads the A rows once per column tile.  A workgroup
// here owns 64 COMPLETE rows (N = 320 or 256 columns = the whole channel axis of the level), so
//   * A is fetched once (not once per column tile), the intermediate row block never leaves the CU: the LayerNorm runs on
//     the accumulators (two-pass, fp32, like norm.hip) and its split32 result is written straight into LDS as the A
//     operand of the second contraction;
//   * only the weights stream: [N lines x 128 B] per 32-deep chunk by LDS-DMA, double / triple buffered.
// Geometry: 256 threads = 4 waves as 2 (rows) x 2 (columns); a wave owns 32 rows x 32 NI columns (NI = 5 -> N = 320,
// NI = 4 -> N = 256): NI accumulators of 32x32 (v_mfma_f32_32x32x16_bf16).  LDS (NI = 5): stage 1 runs three stages of
// [64 A lines | 320 W lines] (144 KB); between the stages the 64 x 320 split32 rows pass through LDS once (the transpose from
// the accumulator layout to the A-operand layout) and every wave keeps its A fragments of all twenty k-steps in registers, so
// stage 2 streams the weights through a FOUR-slot ring over the whole 160 KB -- one workgroup per CU, 195 at M = 12480.
//
// Numerical contract: per accumulator the products are issued k ascending, lo.hi, hi.lo, hi.hi per 16-deep k-step -- the
// order of every other bf16x3 engine, so stage 1 is bit-identical to the unfused GEMM; the LayerNorm sums 2 x 160 columns
// in a fixed tree (results differ from norm.hip's kernel in the last bit, not from run to run).  A row's result depends on
// that row alone: batch invariance holds by construction.
#include "igemm_epilogue.h"

#include <cstdio>
#include <type_traits>

namespace maa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int NT = 256;
constexpr int BM = 64;
constexpr int NS1 = 3;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct RowChainArgs {
    int M;
    // stage 1
    const float* a;
    int lda, K1;
    const float* w1;
    int ldb1;
    const float* bias1;
    const float* res1;
    int ldr1;
    float* y;
    int ldy;
    // between the stages
    const float* ln_g;
    const float* ln_b;
    float eps;
    float* t_out;
    int ldt;
    // stage 2
    const float* w2;
    int ldb2;
    const float* bias2;
    int T2;
    const float* res2;
    int ldr2;
    float* z;
    int ldz, z_split;
    const float* zeros;
};

// bf16 hi / lo words of a value
__device__ __forceinline__ void split1(float v, unsigned& hb, unsigned& lb) {
    const __bf16 h = (__bf16)v;
    hb = __builtin_bit_cast(unsigned short, h);
    const __bf16 l = (__bf16)(v - __builtin_bit_cast(float, hb << 16));
    lb = __builtin_bit_cast(unsigned short, l);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// A wave runs alone on its SIMD (one workgroup per CU: the LDS is full), so nothing hides its own copy issue or fragment
// reads unless they sit BETWEEN its MFMAs: every k-step's matrix instructions carry "fillers" -- the 16-byte fragment reads
// of the next k-step first, then this wave's LDS-DMA pieces of the chunk being prefetched -- one per MFMA, pinned with
// sched_barrier (round 4, after the first version's issue -> read -> multiply sequence measured ~3000 cycles per chunk
// against a 960-cycle matrix floor: profiles/r4_rowchain_ab.txt).
template <int NI, int TERMS>
__global__ __launch_bounds__(NT) void rowchain_kernel(const RowChainArgs q) {
    constexpr int N = 2 * NI * 32;            // columns = the row width of the level
    constexpr int NCH = N / 32;               // 32-deep chunks of a row
    constexpr int ROWS1 = BM + N;             // lines of a stage-1 stage: [A rows | W1 rows]
    constexpr int STAGE1 = ROWS1 * 128;
    constexpr int IPW1 = ROWS1 / 32;          // LDS-DMA pieces (8 lines x 128 B) per wave and chunk
    constexpr int STAGE2 = N * 128;
    constexpr int IPW2 = N / 32;
    constexpr int NS2 = 4;                    // stage-2 weight ring: three chunks in flight
    constexpr int A2_BYTES = BM * N * 4;      // the rows between the stages: [NCH][64 lines][128 B] = two ring slots
    constexpr int SL = NI * TERMS;            // MFMAs (= filler slots) of a k-step
    constexpr int NR = 2 * NI + 2;            // fragment reads of a k-step (stage 1): A hi / lo + NI weight hi / lo
    static_assert(ROWS1 % 32 == 0 && N % 32 == 0, "pieces per wave");
    static_assert(NS1 * STAGE1 <= 163840 && NS2 * STAGE2 <= 163840 && A2_BYTES == 2 * STAGE2, "LDS");
    static_assert(2 * IPW1 <= 63 && 2 * IPW2 <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int lrow = lane & 31, lk = lane >> 5;
    const int r8 = lane >> 3, sl = lane & 7;
    const int m0 = blockIdx.x * BM;
    const char* const zero = reinterpret_cast<const char*>(q.zeros);

    // fragment addressing (tile offsets are multiples of 32 lines: the swizzle depends on lrow only)
    const int swz = (lrow >> 1) & 7;
    int s_off[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) s_off[pl][ks] = ((pl * 4 + ks * 2 + lk) ^ swz) << 4;
    const int a_row = (wm * 32 + lrow) * 128;
    const int b_col = (wn * (NI * 32) + lrow) * 128;

    f32x16 acc[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    struct Frags {
        bf16x8 ah, al, bh[NI], bl[NI];
    };
    // the MFMAs of one k-step: lo.hi, hi.lo, hi.hi per accumulator (the order of every bf16x3 engine); fill(i) after MFMA i
    auto kstep = [&](const bf16x8& ah, const bf16x8& al, const bf16x8 (&bh)[NI], const bf16x8 (&bl)[NI], auto&& fill) __attribute__((always_inline)) {
        static_for<0, SL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int term = TERMS == 3 ? i / NI : 2, j = i % NI;
            if constexpr (term == 0)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[j], 0, 0, 0);
            else if constexpr (term == 1)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[j], 0, 0, 0);
            else
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[j], 0, 0, 0);
            fill(ic);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // fragment read number i of k-step ks (first what the first MFMAs need: A lo, weight hi; then A hi, weight lo)
    auto read_frag = [&](Frags& f, const char* abase, const char* bbase, int ks, auto ic, bool with_a) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i == 0) {
            if (with_a) f.al = *reinterpret_cast<const bf16x8*>(abase + a_row + s_off[1][ks]);
        } else if constexpr (i <= NI) {
            f.bh[i - 1] = *reinterpret_cast<const bf16x8*>(bbase + b_col + (i - 1) * 4096 + s_off[0][ks]);
        } else if constexpr (i == NI + 1) {
            if (with_a) f.ah = *reinterpret_cast<const bf16x8*>(abase + a_row + s_off[0][ks]);
        } else {
            f.bl[i - NI - 2] = *reinterpret_cast<const bf16x8*>(bbase + b_col + (i - NI - 2) * 4096 + s_off[1][ks]);
        }
    };

    // ------------------------------------------------------------------------------------------ stage 1
    {
        // this wave's pieces of a chunk: lines 8 (wid IPW1 + j) + r8 of the stage; A lines first, then W1 lines.  Lane i of
        // a piece lands at base + 16 i, so the 16-byte-slot swizzle is applied to its SOURCE address (as in igemm_dma.hip)
        const char* src[IPW1];
#pragma unroll
        for (int j = 0; j < IPW1; ++j) {
            const int line = 8 * (wid * IPW1 + j) + r8;
            const int slot = sl ^ ((line >> 1) & 7);
            if (line < BM) {
                const long long m = min((long long)m0 + line, (long long)q.M - 1);      // rows past the end: clamped, never stored
                src[j] = reinterpret_cast<const char*>(q.a) + m * q.lda * 4 + slot * 16;
            } else {
                src[j] = reinterpret_cast<const char*>(q.w1) + (long long)(line - BM) * q.ldb1 * 4 + slot * 16;
            }
        }
        const int nch1 = q.K1 / 32;
        int issued = 0;
        // piece j of the next chunk to prefetch (chunks past K: a zero line into the same stage keeps the vmcnt arithmetic uniform)
        auto issue_piece = [&](int stage, auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const bool live = issued < nch1;
            char* dst = smem + stage * STAGE1 + wid * (IPW1 * 1024) + j * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)(live ? src[j] : zero), (lptr_t)dst, 16, 0, 0);
            src[j] += 128;
        };
#pragma unroll
        for (int s = 0; s < NS1 - 1; ++s) {
            static_for<0, IPW1>([&](auto jc) { issue_piece(s, jc); });
            ++issued;
        }
        int st = 0, st_fill = NS1 - 1;
        Frags f0, f1;
        for (int c = 0; c < nch1; ++c) {
            wait_vmcnt<(NS1 - 2) * IPW1>();
            __builtin_amdgcn_s_barrier();          // chunk c is in LDS for everybody; everybody has finished chunk c - 1
            const char* base = smem + st * STAGE1;
            const char* wbase = base + BM * 128;
            static_for<0, NR>([&](auto ic) { read_frag(f0, base, wbase, 0, ic, true); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // k-step 0: its MFMAs carry the reads of k-step 1, then the first pieces of chunk c + 2
            constexpr int D0 = SL > NR ? SL - NR : 0;          // pieces issued under k-step 0
            kstep(f0.ah, f0.al, f0.bh, f0.bl, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < NR)
                    read_frag(f1, base, wbase, 1, ic, true);
                else if constexpr (i - NR < IPW1)
                    issue_piece(st_fill, std::integral_constant<int, i - NR>{});
            });
            static_for<(SL < NR ? SL : NR), NR>([&](auto ic) { read_frag(f1, base, wbase, 1, ic, true); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            kstep(f1.ah, f1.al, f1.bh, f1.bl, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (D0 + i < IPW1) issue_piece(st_fill, std::integral_constant<int, D0 + i>{});
            });
            static_for<(D0 + SL < IPW1 ? D0 + SL : IPW1), IPW1>([&](auto jc) { issue_piece(st_fill, jc); });
            ++issued;
            st = st + 1 == NS1 ? 0 : st + 1;
            st_fill = st_fill + 1 == NS1 ? 0 : st_fill + 1;
        }
        wait_vmcnt<0>();
        __syncthreads();          // every read of the stage-1 ring is done, no copy is in flight: the LDS is re-purposed below
    }

    // ------------------------------------------------------------------------------------------ stage-2 weight stream
    // (t, c) = (column tile, chunk) flattened: the copies never drain between column tiles.  Ring of NS2 = 4 slots over the
    // whole LDS; slots 0 / 1 are its upper half, slots 2 / 3 the lower half, where the rows sit until every wave has taken
    // its A fragments into registers.
    const bool has2 = q.T2 > 0;
    const char* src2[IPW2];
#pragma unroll
    for (int j = 0; j < IPW2; ++j) {
        const int nl = 8 * (wid * IPW2 + j) + r8;
        src2[j] = reinterpret_cast<const char*>(q.w2) + (long long)nl * q.ldb2 * 4 + ((sl ^ ((nl >> 1) & 7)) << 4);
    }
    const int items2 = q.T2 * NCH;
    int issued2 = 0, c_iss = 0;
    const long long tile_step = (long long)N * q.ldb2 * 4 - (long long)NCH * 128;
    auto slot_base = [&](int slot) { return smem + ((slot + 2) & 3) * STAGE2; };
    auto issue2_piece = [&](int slot, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const bool live = issued2 < items2;
        char* dst = slot_base(slot) + wid * (IPW2 * 1024) + j * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(live ? src2[j] : zero), (lptr_t)dst, 16, 0, 0);
        src2[j] += 128;
    };
    auto issue2_done = [&]() __attribute__((always_inline)) {      // bookkeeping after the IPW2 pieces of a chunk
        ++issued2;
        if (++c_iss == NCH) {
            c_iss = 0;
#pragma unroll
            for (int j = 0; j < IPW2; ++j) src2[j] += tile_step;
        }
    };
    if (has2) {          // chunks 0 and 1 land while the rows below are normalised
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            static_for<0, IPW2>([&](auto jc) { issue2_piece(s, jc); });
            issue2_done();
        }
    }

    // ------------------------------------------------------------------------------------------ stage-1 epilogue
    // accumulator layout: lane holds column lrow of each 32-wide block, rows (r & 3) + 8 (r >> 2) + 4 lk
    const int rowt = wm * 32 + 4 * lk;            // + (r & 3) + 8 (r >> 2): row inside the tile
    long long mc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + rowt + (r & 3) + 8 * (r >> 2);
        mc[r] = m < q.M ? m : q.M - 1;
    }
    {
        float bias[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) bias[j] = q.bias1 ? q.bias1[wn * (NI * 32) + j * 32 + lrow] : 0.f;
        if (q.res1) {
            // all of the residual's loads in flight together, waited for once (igemm_epilogue.h explains the cost otherwise)
            float rs[NI][16];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) rs[j][r] = q.res1[mc[r] * q.ldr1 + wn * (NI * 32) + j * 32 + lrow];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                settle(bias[j]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    settle(rs[j][r]);
                    acc[j][r] = (acc[j][r] * 1.0f + bias[j]) + rs[j][r];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                settle(bias[j]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] * 1.0f + bias[j];
            }
        }
    }
    if (q.y) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + rowt + (r & 3) + 8 * (r >> 2);
                if (m < q.M) q.y[(long long)m * q.ldy + wn * (NI * 32) + j * 32 + lrow] = acc[j][r];
            }
    }

    // ---- LayerNorm over the N columns of every row (two passes, fp32): a wave holds 32 NI of a row's columns -- block sum
    // in the lane, tree over the 32 lanes of its half, then the partner wave's half through LDS
    if (q.ln_g) {
        float* red = reinterpret_cast<float*>(smem);          // [2 passes][2 wn][64 rows] (inside the rows' region, not yet written)
        float mean[16], rstd[16];
        {
            float s[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[0][r];
#pragma unroll
                for (int j = 1; j < NI; ++j) t += acc[j][r];
                s[r] = t;
            }
#pragma unroll
            for (int o = 1; o < 32; o <<= 1)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += __shfl_xor(s[r], o, 64);
            if (lrow == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wn * 64 + rowt + (r & 3) + 8 * (r >> 2)] = s[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rowt + (r & 3) + 8 * (r >> 2);
                mean[r] = (red[row] + red[64 + row]) / (float)N;
            }
        }
        {
            float s[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const float d = acc[j][r] - mean[r];
                    t += d * d;
                }
                s[r] = t;
            }
#pragma unroll
            for (int o = 1; o < 32; o <<= 1)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += __shfl_xor(s[r], o, 64);
            if (lrow == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[128 + wn * 64 + rowt + (r & 3) + 8 * (r >> 2)] = s[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rowt + (r & 3) + 8 * (r >> 2);
                rstd[r] = 1.f / sqrtf((red[128 + row] + red[192 + row]) / (float)N + q.eps);
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = wn * (NI * 32) + j * 32 + lrow;
            const float g = q.ln_g[n], b = q.ln_b[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = (acc[j][r] - mean[r]) * rstd[r] * g + b;
        }
        __syncthreads();          // every wave has read the partial sums: the region may now take the rows
    }

    // ---- t as split32: into LDS (chunk-major, swizzled like a staged tile) for stage 2, and / or to global memory.
    // The two lanes of an even / odd column pair swap one half each (store_split_pair's trick): one 4-byte store per lane.
    {
        const bool even = (lrow & 1) == 0;
        const int boff = even ? lrow * 2 : 64 + (lrow - 1) * 2;          // byte inside the 128-byte line: hi plane | lo plane
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int chunk = wn * NI + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rowt + (r & 3) + 8 * (r >> 2);
                unsigned hb, lb;
                split1(acc[j][r], hb, lb);
                const unsigned mine = even ? lb : hb;
                const unsigned theirs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);
                const unsigned word = even ? (hb | (theirs << 16)) : (theirs | (lb << 16));
                if (has2) {
                    const int slot = (boff >> 4) ^ ((row >> 1) & 7);
                    *reinterpret_cast<unsigned*>(smem + (chunk * 64 + row) * 128 + (slot << 4) + (boff & 15)) = word;
                }
                if (q.t_out) {
                    const int m = m0 + row;
                    if (m < q.M)
                        *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(q.t_out + (long long)m * q.ldt) + chunk * 128 + boff) = word;
                }
            }
        }
    }
    if (!has2) return;

    // ------------------------------------------------------------------------------------------ stage 2
    // the rows' A fragments of every k-step into registers (a wave runs alone on its SIMD: 512 registers), then the lower half
    // of the LDS joins the weight ring
    bf16x8 a2h[NCH][2], a2l[NCH][2];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a2h[c][ks] = *reinterpret_cast<const bf16x8*>(smem + c * (64 * 128) + a_row + s_off[0][ks]);
            if constexpr (TERMS == 3) a2l[c][ks] = *reinterpret_cast<const bf16x8*>(smem + c * (64 * 128) + a_row + s_off[1][ks]);
        }
    __syncthreads();
    static_for<0, IPW2>([&](auto jc) { issue2_piece(2, jc); });
    issue2_done();

#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int slot = 0, slot_fill = NS2 - 1;
    Frags g0, g1;
    for (int t = 0; t < q.T2; ++t) {
        static_for<0, NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            wait_vmcnt<(NS2 - 2) * IPW2>();          // this wave's pieces of the chunk have landed (two younger chunks may be in flight)
            __builtin_amdgcn_s_barrier();            // everybody's have; everybody has finished the previous chunk
            const char* wbase = slot_base(slot);
            static_for<0, NR>([&](auto ic) { read_frag(g0, nullptr, wbase, 0, ic, false); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            constexpr int D0 = SL > NR ? SL - NR : 0;
            kstep(a2h[c][0], a2l[c][0], g0.bh, g0.bl, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < NR)
                    read_frag(g1, nullptr, wbase, 1, ic, false);
                else if constexpr (i - NR < IPW2)
                    issue2_piece(slot_fill, std::integral_constant<int, i - NR>{});
            });
            static_for<(SL < NR ? SL : NR), NR>([&](auto ic) { read_frag(g1, nullptr, wbase, 1, ic, false); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            kstep(a2h[c][1], a2l[c][1], g1.bh, g1.bl, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (D0 + i < IPW2) issue2_piece(slot_fill, std::integral_constant<int, D0 + i>{});
            });
            static_for<(D0 + SL < IPW2 ? D0 + SL : IPW2), IPW2>([&](auto jc) { issue2_piece(slot_fill, jc); });
            issue2_done();
            slot = (slot + 1) & 3;
            slot_fill = (slot_fill + 1) & 3;
        });
        // ---- epilogue of column tile t (the next tile's first chunks are already on their way)
        float bias[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) bias[j] = q.bias2 ? q.bias2[t * N + wn * (NI * 32) + j * 32 + lrow] : 0.f;
        if (q.res2) {
            float rs[NI][16];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) rs[j][r] = q.res2[mc[r] * q.ldr2 + t * N + wn * (NI * 32) + j * 32 + lrow];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                settle(bias[j]);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    settle(rs[j][r]);
                    acc[j][r] = (acc[j][r] * 1.0f + bias[j]) + rs[j][r];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                settle(bias[j]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = acc[j][r] * 1.0f + bias[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = t * N + wn * (NI * 32) + j * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + rowt + (r & 3) + 8 * (r >> 2);
                if (m < q.M) {
                    if (q.z_split)
                        store_split_pair(q.z + (long long)m * q.ldz, n, acc[j][r]);
                    else
                        q.z[(long long)m * q.ldz + n] = acc[j][r];
                }
                acc[j][r] = 0.f;
            }
        }
    }
    wait_vmcnt<0>();          // (the trailing dummy copies) nothing may land in LDS after the workgroup has given it back
}

template <int NI, int TERMS>
void launch_one(const Ctx& ctx, const RowChainArgs& q) {
    constexpr int N = 2 * NI * 32;
    constexpr size_t l1 = (size_t)NS1 * (BM + N) * 128, l2 = 4 * (size_t)N * 128;
    constexpr size_t lds = l1 > l2 ? l1 : l2;
    auto kern = rowchain_kernel<NI, TERMS>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((q.M + BM - 1) / BM)), dim3(NT), lds, ctx.stream, q);
}

}  // namespace

bool rowchain_covers(const Ctx& ctx, const RowChain& d) {
    if (ctx.dtype == 0) return false;      // (whether the UNet USES the chains is the caller's policy: Tuning::rowchain, unet.cpp)
    const int N = d.w1.N;
    if (N != 320 && N != 256) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!(d.w1.split && d.w1.nk && d.w1.K % 32 == 0 && d.w1.K >= 32 && d.w1.ld % 32 == 0 && d.w1.Npad >= N && al16(d.w1.w))) return false;
    if (!(d.lda % 32 == 0 && d.lda >= d.w1.K && al16(d.a))) return false;
    if (d.w2) {
        const PackedW& w2 = *d.w2;
        if (!(w2.split && w2.nk && w2.K == N && w2.N % N == 0 && w2.ld % 32 == 0 && w2.Npad >= w2.N && al16(w2.w))) return false;
        if (d.res2 && w2.N != N) return false;
        if (d.z_split && (d.ldz % 32 != 0 || !al16(d.z))) return false;
    }
    if (d.t_out && (d.ldt % 32 != 0 || !al16(d.t_out))) return false;
    return d.M > 0;
}

void launch_rowchain(const Ctx& ctx, const RowChain& d) {
    MAA_CHECK(rowchain_covers(ctx, d), "rowchain: problem not covered");
    if (ctx.ws.dry) return;
    const int N = d.w1.N;
    RowChainArgs q;
    q.M = d.M;
    q.a = d.a;
    q.lda = d.lda;
    q.K1 = d.w1.K;
    q.w1 = d.w1.w;
    q.ldb1 = d.w1.ld;
    q.bias1 = d.w1.bias;
    q.res1 = d.res1;
    q.ldr1 = d.ldr1;
    q.y = d.y;
    q.ldy = d.ldy;
    q.ln_g = d.ln_g;
    q.ln_b = d.ln_b;
    q.eps = d.eps;
    q.t_out = d.t_out;
    q.ldt = d.ldt;
    q.w2 = d.w2 ? d.w2->w : nullptr;
    q.ldb2 = d.w2 ? d.w2->ld : 0;
    q.bias2 = d.w2 ? d.w2->bias : nullptr;
    q.T2 = d.w2 ? d.w2->N / N : 0;
    q.res2 = d.res2;
    q.ldr2 = d.ldr2;
    q.z = d.z;
    q.ldz = d.ldz;
    q.z_split = d.z_split;
    q.zeros = ctx.zeros;
    MAA_CHECK(!d.w2 || d.z, "rowchain: stage 2 needs its output");
    const double flops = 2.0 * d.M * (double)N * d.w1.K + (d.w2 ? 2.0 * d.M * (double)d.w2->N * N : 0.0);
    const double bytes = 4.0 * ((double)d.M * d.w1.K + (double)N * d.w1.K + (d.w2 ? (double)d.w2->N * (N + d.M) : 0.0) +
                                (d.y ? (double)d.M * N : 0.0) + (d.res1 ? (double)d.M * N : 0.0));
    char shape[64];
    const char* name = N == 320 ? "igemm_rowchain_bf16x3<64x320>" : "igemm_rowchain_bf16x3<64x256>";
    if (ctx.dtype == 2) name = N == 320 ? "igemm_rowchain_bf16<64x320>" : "igemm_rowchain_bf16<64x256>";
    if (ctx.prof && ctx.prof->detail) {
        std::snprintf(shape, sizeof(shape), "rc M%d N%d K%d N2 %d ln%d", d.M, N, d.w1.K, d.w2 ? d.w2->N : 0, d.ln_g ? 1 : 0);
        name = shape;
    }
    ProfScope prof(ctx, name, flops, bytes);
    if (ctx.dtype == 1) {
        if (N == 320)
            launch_one<5, 3>(ctx, q);
        else
            launch_one<4, 3>(ctx, q);
    } else {
        if (N == 320)
            launch_one<5, 1>(ctx, q);
        else
            launch_one<4, 1>(ctx, q);
    }
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
