// bf16x3 implicit GEMM, third LDS-DMA engine: the UNet's 3x3 convolutions with a HALO-STAGED A operand and two wave groups
// per workgroup that alternate between a matrix phase and a memory phase ("ping-pong").
//
// Why (DESIGN.md 3.2 / VERDICT r2 #1, #2).  The second engine (igemm_dma2.hip) runs its K loop over (tap, channel chunk) and
// DMAs every A line nine times, once per tap; all of a workgroup's waves issue their copies, read their fragments and
// multiply in the SAME phase between the same barriers, so a SIMD's matrix pipe idles whenever its single wave issues LDS-DMA
// pieces (60-180 cycles each) or waits for fragments: 40 % MFMA-busy.  Here
//   * the K loop is (channel chunk outer, 9 taps inner).  The input positions an M tile of 256 outputs can touch -- the flat
//     range [m0 - W - 1, m0 + 255 + W + 1] of the channels-last activation -- are staged ONCE per 32-channel chunk
//     (256 + 2 W + 2 split32 lines) and every tap reads them at a line offset ky W + kx; lanes whose tap falls outside the
//     image (row / sample edges of the flat range) read a zero line instead.  A's LDS-DMA bytes per (chunk x 9 taps) drop
//     from 9 x 32 KB to 42-52 KB and its fabric re-reads from 9x to ~1.6x; only the weights stream tap by tap.
//   * a workgroup is 8 waves = two groups of four; group g owns rows [128 g, 128 g + 128) of the 256 x BN tile.  Every SIMD
//     hosts one wave of each group (waves w and w + 4 share a SIMD).  Time is cut into ticks separated by s_barrier: in a
//     tick one group multiplies chunk q out of registers (2 k-steps x 3 MI NI MFMAs, nothing else) while the other reads
//     ITS fragments of the chunk it multiplies next and issues its share of the LDS-DMA pieces; then they swap.  The
//     matrix pipe of a SIMD always has one wave feeding it and the partner's memory phase costs it nothing
//     (MI355X_MICROARCH.md, "Two waves per SIMD": matrix beside memory is the complementary pairing).
//
// Tick t: group g reads chunk q at tick 2 q + g and multiplies it at tick 2 q + g + 1.  The weights of chunk q live in slot
// q % 3 of a 3-slot ring and are needed during ticks 2 q .. 2 q + 1; the pieces of chunk q + 2 are issued into the slot chunk
// q - 1 has left, half by group 0 at tick 2 q, half by group 1 at tick 2 q + 1, and every wave ends a memory phase with
// s_waitcnt vmcnt(NP) -- everything but the NP pieces it has just issued has landed -- so a piece has two ticks (~1600
// cycles) to arrive.  A is double-buffered by channel chunk: the pieces of chunk c + 1 are issued one per wave and memory
// phase over taps 0..7 of chunk c.  Every wave issues exactly NP pieces per memory phase (pieces that do not exist copy the
// zero page into a dump line) so the vmcnt arithmetic is the same constant everywhere.
//
// Numerical contract: per accumulator the products of a k-step are issued lo.hi, hi.lo, hi.hi like the other engines, but
// the k-steps run (channel chunk, tap, k) instead of (tap, channel, k): results differ from the other engines in the last
// bits; which engine a layer takes is a function of the layer (taps, W, C, N), never of M, and so is the number of K
// slices -- a sample's result does not depend on its batch.
#include "igemm_epilogue.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace maa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;          // channels per chunk = one split32 line
constexpr int BM = 256;         // output rows per workgroup
constexpr int NSB = 3;          // weight ring

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct PPArgs {
    int W, H;            // image width / height (stride 1, "same" padding: output = input geometry; 1-D: H = 1, W = L)
    int T, KW;           // taps, taps per kernel row (tap t = (ky, kx) = (t / KW, t % KW))
    int rowstep, colstep;// lines between the kernel's rows / columns in the flat channels-last activation: dh W, dw
    int dh, dw, ph, pw;  // dilation and padding (the edge masks)
    int padflat;         // the tile's first line is flat position m0 - padflat (= ph W + pw)
    int NLp;             // A lines of one channel chunk: 256 + the last tap's offset, rounded up to a multiple of 16
    int CAPl;            // lines of the A ring (a multiple of 16, NLp + the host's margin .. 2 NLp)
    int ntiles, tiles;   // N tiles, M tiles x N tiles
    int items;           // tiles x K slices
    int nci, cps;        // channel chunks in all, per K slice
    int Nb;              // rows of the packed weight that exist
    float* part;         // split-K slabs or null
    int S, tile_major;   // K slices; 1: the slices of a tile are neighbours in the item order (one XCD writes a tile's slabs, the XCD
                         // whose reduce blocks read them and whose rows the consumer reads: xcd_contiguous, maa_internal.h), 0: slice-major
    int tiles_pp;        // UP2: tiles of ONE phase (items = 4 phases x tiles_pp); phase = (py, px) of the 2x-upsampled output pixel
    long long b_phase;   // UP2: floats between the packed weights of consecutive phases
    int dbg;             // TUNE instantiation only (MAA_PP_DBG): 1 no MFMAs, 2 no copies after the prologue, 4 no fragment reads, 8 no vmcnt wait
};

// MI x NI fragments of 32x32 per wave; a group's 128 x BN block is GWM x GWN waves (GWM GWN = 4, GWM 32 MI = 128)
// NPA: A pieces a wave issues per memory phase (1 for 3x3 / long 1-D kernels, 3 for the 3-tap 1-D kernels whose next chunk
// has only two taps' worth of phases to arrive in)
// PERSIST = false: a launch whose grid covers every item (the UNet's: 196-208 items on 256 CUs) -- the epilogue does not carry
// the next item's staging state, which is what made the 256 x 160 instantiation spill 81 VGPRs
// OUT: 0 = the result leaves through the fused epilogue or as a split-K slab (run-time choice; the TUNE build), 1 = epilogue
// only, 2 = slab only -- the UNet's launches are all of the last kind, and an instantiation without the epilogue's registers
// has no scratch at all
// TERMS: 3 = bf16x3; 1 = the context's plain-bf16 mode: the hi halves of the same split32 lines are the operands (one MFMA per
// k-step, the lo halves are staged with their lines but never read) -- the same schedule, a third of the matrix phase
// UP2: "nearest-2x upsample, then 3x3 convolution" (openaimodel.py:116-118, model.py:52-56) as FOUR 2x2 convolutions of the
// low-resolution input, one per parity (py, px) of the output pixel: the 3x3 taps that read the same source pixel are summed into one
// weight at load (runtime.cpp pack_conv_up2) -- 4 / 9 of the multiplications.  One launch: item -> (phase, tile); a phase has its
// own packed weights (b + phase * b_phase), padding (ph, pw) = (1 - py, 1 - px) and output plane (c + phase * M * ldc, interleaved
// into the image by pixel_shuffle2_kernel afterwards).  No K split (OUT = 1).
template <int MI, int NI, int GWM, int GWN, int NPA, bool TUNE, bool PERSIST, int OUT, int TERMS, bool UP2 = false>
__global__ __launch_bounds__(512) void igemm_pp_kernel(const IGemm p, const PPArgs q) {
    constexpr int BN = GWN * NI * 32;
    constexpr int BPG = BN / 16;                    // weight pieces (8 rows x 128 B) per group and chunk: half a chunk
    constexpr int NPB = (BPG + 3) / 4;              // ... per wave and memory phase
    constexpr int NP = NPB + NPA;                   // + the A pieces
    static_assert(GWM * GWN == 4 && GWM * MI * 32 == 128, "group geometry");
    static_assert(BN % 16 == 0, "weight pieces");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // [A buffer 0][A buffer 1][weight slot 0..2][zero line 128 B][dump 1 KB]
    const int a_bytes = q.CAPl * 128;      // the whole A ring
    char* const sA = smem;
    char* const sB = smem + a_bytes;
    char* const sZ = sB + NSB * BN * 128;
    char* const sD = sZ + 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wq = wid & 3;
    const int wm = wq / GWN, wn = wq - wm * GWN;
    const int lrow = lane & 31, lk = lane >> 5;

    // ---- work items: (K slice, M tile, N tile), q.items of them; an XCD (block b runs on XCD b % 8) owns a contiguous range
    // and its workgroups walk it with a stride of their number.  A workgroup is persistent (the grid is capped at one per CU:
    // the LDS holds one): it issues the first copies of its next item before the epilogue of the current one, so the cold
    // start of an item hides under the stores of its predecessor (the vocoders' layers are 16 384 items of 12-44 chunks).
    int w_lo, w_cnt, w_step;
    {
        const int G = (int)gridDim.x, xcd = blockIdx.x & 7, qq = q.items >> 3, rr = q.items & 7;
        w_lo = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
        w_cnt = qq + (xcd < rr ? 1 : 0);
        w_step = (G - xcd + 7) >> 3;
    }
    int w_cur = (int)(blockIdx.x >> 3);
    if (w_cur >= w_cnt) return;              // (never when grid <= items; uniform for the workgroup)
    const int TAPS = q.T;
    const int W = q.W, H = q.H;
    const long long Mtot = p.M;
    int item = 0, m0 = 0, n0 = 0, c_begin = 0, c_end = 0, NQ = 0, cur_slab = 0;
    int phase = 0, ph_i = q.ph, pw_i = q.pw, padflat_i = q.padflat;      // (UP2: per item; else the launch's)
    const char* b_i = reinterpret_cast<const char*>(p.b);

    // zero line (read by lanes whose tap is outside the image)
    if (tid < 8) *reinterpret_cast<f32x4*>(sZ + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing
    // A: row r of the tile sits at line r + ky rowstep + kx colstep of the chunk for tap (ky, kx); 16-byte slot s of line l is
    // stored at slot s ^ ((l >> 1) & 7).  valid9: bit t set when tap t of this lane's row is inside the image.
    int a_r[MI];
    unsigned valid9[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a_r[i] = grp * 128 + wm * (32 * MI) + i * 32 + lrow;
    const int lk16 = lk << 4;
    const unsigned zaddr = (unsigned)(sZ - smem);
    // B: row n of the slot at n * 128, slot swizzle by the row (tile offsets are multiples of 32 rows)
    const int b_row = (wn * (32 * NI) + lrow) * 128;
    const int b_swz = (lrow >> 1) & 7;
    int b_off[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b_off[pl][ks] = b_row + (((pl * 4 + ks * 2 + lk) ^ b_swz) << 4);

    // ---- copies.  Lane i of a piece moves the 16 bytes of slot (i & 7) ^ swizzle(line) of line 8 piece + (i >> 3).
    // Every source address is valid memory whatever happens to its destination: weight rows and A positions are clamped
    // (a clamped A line is only ever read by lanes whose tap is masked), pieces that do not exist land in the dump line.
    const int r8 = lane >> 3, sl = lane & 7;
    // weights: this wave's pieces of a chunk are pb = grp BPG + k 4 + wq (k < NPB, existing while k 4 + wq < BPG)
    const char* gpb[NPB];
    const int C4 = p.C1 * 4;                           // bytes between the taps of a weight row
    // A: during tap t of channel chunk c this wave issues pieces ((2 t + grp) 4 + wq) NPA + i of chunk c + 1
    const char* const a_base = reinterpret_cast<const char*>(p.a1);
    const unsigned lda4 = (unsigned)p.lda1 * 4u;      // bytes between positions (< 2^31)
    const int Mlast = p.M - 1;
    int a_P0 = 0;                                     // flat position of this lane's line of piece 0
    const int a_pieces = q.NLp >> 3;
    // per-item state: tile origin, K range, edge masks, weight row pointers
    auto setup_item = [&](int w) __attribute__((always_inline)) {
        item = w_lo + w;
        // (the quotients are wave-uniform but come out of the VALU's division sequence: back into SGPRs)
        const int slice = __builtin_amdgcn_readfirstlane(q.tile_major ? item % q.S : item / q.tiles);
        int tile = __builtin_amdgcn_readfirstlane(q.tile_major ? item / q.S : item - slice * q.tiles);
        cur_slab = slice * q.tiles + tile;
        if constexpr (UP2) {
            phase = __builtin_amdgcn_readfirstlane(tile / q.tiles_pp);
            tile -= phase * q.tiles_pp;
            ph_i = 1 - (phase >> 1);
            pw_i = 1 - (phase & 1);
            padflat_i = ph_i * W + pw_i;
            b_i = reinterpret_cast<const char*>(p.b) + (long long)phase * q.b_phase * 4;
        }
        const int mt = __builtin_amdgcn_readfirstlane(tile / q.ntiles), nt = tile - mt * q.ntiles;
        m0 = mt * BM;
        n0 = nt * BN;
        c_begin = slice * q.cps;
        c_end = min(q.nci, c_begin + q.cps);
        NQ = (c_end - c_begin) * TAPS;
        a_P0 = m0 - padflat_i + r8;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const long long m = (long long)m0 + a_r[i];
            unsigned v = 0;
            if (m < Mtot) {
                const int ox = (int)(m % W), oy = (int)((m / W) % H);
                for (int t = 0; t < TAPS; ++t) {
                    const int iy = oy + (t / q.KW) * q.dh - ph_i, ix = ox + (t % q.KW) * q.dw - pw_i;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v |= 1u << t;
                }
            }
            valid9[i] = v;
        }
#pragma unroll
        for (int k = 0; k < NPB; ++k) {
            const int pb = grp * BPG + k * 4 + wq;
            const int nl = 8 * pb + r8;
            const int n = min(n0 + nl, q.Nb - 1);          // rows past the last one: clamped, their columns are never stored
            gpb[k] = b_i + (long long)n * p.ldb * 4 + ((sl ^ ((nl >> 1) & 7)) << 4);
        }
    };
    // any piece (prologue)
    auto issue_a_any = [&](int ci, int pa) __attribute__((always_inline)) {
        const int line = pa * 8 + r8;
        const int P = min(max(m0 - padflat_i + line, 0), Mlast);
        const char* src = a_base + ((unsigned long long)(unsigned)P * lda4 + (unsigned long long)ci * 128u) + ((sl ^ ((line >> 1) & 7)) << 4);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + pa * 1024), 16, 0, 0);      // (the first chunk starts at line 0)
    };

    // first copies of the current item: A of its first chunk and the weights of chunks 0 and 1, by all eight waves
    auto prologue = [&]() __attribute__((always_inline)) {
        for (int pa = wid; pa < a_pieces; pa += 8) issue_a_any(c_begin, pa);
#pragma unroll
        for (int j = 0; j < NSB - 1; ++j) {
            // (an item has at least T >= 3 chunks)
            for (int pb = wid; pb < BN / 8; pb += 8) {
                const int nl = 8 * pb + r8;
                const int n = min(n0 + nl, q.Nb - 1);
                const char* src = b_i + (long long)n * p.ldb * 4 + ((sl ^ ((nl >> 1) & 7)) << 4) +
                                  ((long long)j * C4 + (long long)c_begin * 128);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sB + j * (BN * 128) + pb * 1024), 16, 0, 0);
            }
        }
    };

    f32x16 acc[MI][NI];

    bf16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];      // [k-step][fragment]

    // loop state (all wave-uniform): the chunk being read is (ci, t) in ring slot `slot`, its tap offset in lines `shift`;
    // the chunk whose weights are issued is two ahead: byte offset boff2 in a weight row, slot slot2, live while j + 2 < NQ
    // A ring: the chunk being read starts at line a_o, the next one at piece a_on8 (lines a_o + NLp, wrapped)
    const int cap8 = q.CAPl >> 3;
    int a_o = 0, a_on8 = 0;
    int ci = 0, t = 0, kx = 0, shift = 0, slot = 0;
    int slot2 = 0, t2 = 0;
    long long boff2 = 0;

    // memory phase of chunk j = (ci, t): fragments of the chunk into registers, this wave's pieces of chunk j + 2 and of
    // the next channel chunk's A on their way
    auto reads = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // line of the ring: the chunk starts at a_o and wraps at CAPl (unsigned min picks the wrapped value when it exists)
            const unsigned l0 = (unsigned)(a_o + shift) + (unsigned)a_r[i];
            const unsigned line = min(l0, l0 - (unsigned)q.CAPl);
            const unsigned t1 = (unsigned)lk16 ^ ((line << 3) & 0x70u);      // lk slot bit ^ swizzle((line >> 1) & 7)
            const bool ok = (valid9[i] >> t) & 1u;
            // a masked lane reads the zero line: its four 16-byte pieces at zaddr + (0 .. 0x70) are all inside it
            const unsigned base = ok ? line * 128u : zaddr;
            ah[0][i] = *reinterpret_cast<const bf16x8*>(smem + (base + (0x00u ^ t1)));
            ah[1][i] = *reinterpret_cast<const bf16x8*>(smem + (base + (0x20u ^ t1)));
            if constexpr (TERMS == 3) {
                al[0][i] = *reinterpret_cast<const bf16x8*>(smem + (base + (0x40u ^ t1)));
                al[1][i] = *reinterpret_cast<const bf16x8*>(smem + (base + (0x60u ^ t1)));
            }
        }
        const char* const bs = sB + slot * (BN * 128);
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bh[ks][jn] = *reinterpret_cast<const bf16x8*>(bs + jn * 4096 + b_off[0][ks]);
                if constexpr (TERMS == 3) bl[ks][jn] = *reinterpret_cast<const bf16x8*>(bs + jn * 4096 + b_off[1][ks]);
            }
        }
    };
    auto copies = [&](int j) __attribute__((always_inline)) {
        // weights of chunk j + 2 -> the slot chunk j - 1 has left (both groups have read it: the barrier before this tick)
        const bool live = j + (NSB - 1) < NQ;
        static_for<0, NPB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const bool real = live && (k * 4 + wq < BPG);       // wave-uniform
            char* dst = real ? sB + slot2 * (BN * 128) + (grp * BPG + k * 4 + wq) * 1024 : sD;
            __builtin_amdgcn_global_load_lds((gptr_t)(gpb[k] + boff2), (lptr_t)dst, 16, 0, 0);
        });
        // A of the next channel chunk: pieces ((2 t + grp) 4 + wq) NPA + i during every tap but the last
        static_for<0, NPA>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int pa = ((t * 2 + grp) * 4 + wq) * NPA + i;
            const bool real = t < TAPS - 1 && ci + 1 < c_end && pa < a_pieces;
            const int P = min(max(a_P0 + 8 * pa, 0), Mlast);
            // (the ring and a chunk hold an even number of pieces: the swizzle parity of a ring piece is that of pa)
            const unsigned swz16 = (unsigned)((sl ^ (((r8 >> 1) + 4 * (pa & 1)) & 7)) << 4);
            const char* src = a_base + ((unsigned long long)(ci + 1) * 128u) + ((unsigned long long)(unsigned)P * lda4 + swz16);
            int pp = a_on8 + pa;
            if (pp >= cap8) pp -= cap8;
            char* dst = real ? sA + pp * 1024 : sD;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
        });
    };
    // memory phase of chunk j = (ci, t): the fragments of the chunk into registers, then this wave's pieces of chunk j + 2 and
    // of the next channel chunk's A on their way.  (Measured, profiles/r3_pp_ablate_v3_order_variants.txt: copies before the
    // reads, or the fragments waited for only at the head of the matrix phase, are both 25-30 % slower; s_setprio on either
    // phase changes nothing.)
    auto load_phase = [&](int j) __attribute__((always_inline)) {
        if (!TUNE || !(q.dbg & 4)) reads();
        if (!TUNE || !(q.dbg & 2)) copies(j);
        wait_lgkm0();
        if (!TUNE || !(q.dbg & 8)) wait_vmcnt<NP>();       // all but the NP pieces just issued have landed
    };
    auto advance = [&]() __attribute__((always_inline)) {
        shift += q.colstep;
        if (++kx == q.KW) {
            kx = 0;
            shift += q.rowstep - q.KW * q.colstep;
        }
        if (++t == TAPS) {
            t = 0;
            shift = 0;
            ++ci;
            a_o = a_on8 * 8;
            a_on8 += a_pieces;
            if (a_on8 >= cap8) a_on8 -= cap8;
        }
        if (++slot == NSB) slot = 0;
        if (++slot2 == NSB) slot2 = 0;
        boff2 += C4;
        if (++t2 == TAPS) {
            t2 = 0;
            boff2 += 128 - (long long)TAPS * C4;
        }
    };

    auto mma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (TERMS == 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][jn], acc[i][jn], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][jn], acc[i][jn], 0, 0, 0);
        }
    };

    setup_item(w_cur);
    prologue();
    for (;;) {
        wait_vmcnt<0>();
        __syncthreads();                 // this item's first chunks are in LDS (and, first time, the zero line)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        a_o = 0;
        a_on8 = a_pieces;                // (the ring is longer than a chunk)
        ci = c_begin;
        t = kx = shift = slot = 0;
        slot2 = t2 = NSB - 1;
        boff2 = (long long)(NSB - 1) * C4 + (long long)c_begin * 128;

        // ---- ticks.  Group 1 runs one tick behind group 0; every wave passes 2 NQ barriers.
        if (grp == 1) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int j = 0; j < NQ; ++j) {
            load_phase(j);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (!TUNE || !(q.dbg & 1)) mma_phase();
            __builtin_amdgcn_sched_barrier(0);
            if (!(grp == 1 && j == NQ - 1)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            advance();
        }
        // Every fragment read of this item is done (a wave gets here through the barrier that follows group 1's last memory
        // phase): the next item's first copies may overwrite the rings while this item's results are stored.
        const int e_slab = cur_slab, e_m0 = m0, e_n0 = n0, e_phase = phase;
        const int w_next = w_cur + w_step;
        bool more = false;
        if constexpr (PERSIST) {
            more = w_next < w_cnt;
            if (more) {
                setup_item(w_next);
                prologue();
            }
        }

        // ---- epilogue or slab
        const int rpb = p.Hout * p.Wout;
        const int row_base = e_m0 + grp * 128 + wm * (32 * MI), col_base = e_n0 + wn * (32 * NI);
        if (OUT == 1 || (OUT == 0 && q.part == nullptr)) {
            igemm_epilogue<MI, NI>(p, acc, row_base, col_base, lrow, lk, UP2 ? (long long)e_phase * p.M * p.ldc : 0LL, q.Nb, rpb);
        } else {
            // slab of this (slice, tile): [MI NI blocks][4 register quads][512 threads][4 floats]
            float* pp = q.part + ((long long)e_slab * (MI * NI * 4) * 512 + tid) * 4;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const f32x4 v = {acc[i][jn][4 * qd], acc[i][jn][4 * qd + 1], acc[i][jn][4 * qd + 2], acc[i][jn][4 * qd + 3]};
                        *reinterpret_cast<f32x4*>(pp + (long long)((i * NI + jn) * 4 + qd) * 512 * 4) = v;
                    }
        }
        if (!more) break;
        w_cur = w_next;
    }
    wait_vmcnt<0>();        // (dummies only) nothing may land in LDS after the workgroup has given it back
}

// Tap geometry of a problem this engine takes (igemm_pp_plan checks eligibility)
struct PPGeom {
    int T, KW, rowstep, colstep, padflat, NLp, npa;
};
PPGeom pp_geom(const IGemm& p) {
    PPGeom g;
    g.T = p.KH * p.KW;
    g.KW = p.KW;
    g.rowstep = p.dh * p.Win;
    g.colstep = p.dw;
    g.padflat = p.ph * p.Win + p.pw;
    // lines one channel chunk of A occupies: 256 rows + the last tap's offset, a multiple of 16 (an even number of pieces:
    // see the swizzle parity in the kernel)
    g.NLp = (BM + (p.KH - 1) * g.rowstep + (p.KW - 1) * g.colstep + 15) / 16 * 16;
    // pieces per wave and memory phase so that the next chunk arrives during all taps but the last: the kernel is
    // instantiated for 1 and 3 (0: more than that, not taken)
    const int need = (g.NLp / 8 + 8 * (g.T - 1) - 1) / (8 * (g.T - 1));
    g.npa = need <= 1 ? 1 : need <= 3 ? 3 : 0;
    return g;
}
// Lines of the A ring beside a weight ring of NSB x bn lines, 0 when it does not fit.  Two whole chunks (plain double
// buffering) when there is room; else the next chunk's pieces wrap into the lines the current one no longer needs: piece pa is
// issued during tap t = pa / (8 npa) and lands on the current chunk's lines NLp + 8 pa - CAP .. + 7, which must lie below the
// first line tap t = (ky, kx) and every later tap read, ky rowstep + kx colstep (the tap offsets ascend with t).
int pp_ring_lines(const PPGeom& g, int bn) {
    const int room = (163840 - 1152 - NSB * bn * 128) / 128 / 16 * 16;
    if (room >= 2 * g.NLp) return 2 * g.NLp;
    int X = 0;
    for (int pa = 0; pa < g.NLp / 8; ++pa) {
        const int t = pa / (8 * g.npa);
        if (t >= g.T - 1) return 0;
        const int need = 8 * pa + 8 - ((t / g.KW) * g.rowstep + (t % g.KW) * g.colstep);
        if (need > X) X = need;
    }
    const int cap = g.NLp + (X + 15) / 16 * 16;
    return cap <= room ? cap : 0;
}

template <int MI, int NI, int GWM, int GWN, int NPA, int TERMS>
void launch_npa_terms(const Ctx& ctx, const IGemm& p, const PPArgs& q, int items, size_t lds) {
    auto go = [&](auto kern) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, 163840);
        hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(512), lds, ctx.stream, p, q);
    };
    const bool persist = q.items > items, slab = q.part != nullptr;      // (more items than workgroups: persistent)
    if (persist && slab) go(igemm_pp_kernel<MI, NI, GWM, GWN, NPA, false, true, 2, TERMS>);
    else if (persist) go(igemm_pp_kernel<MI, NI, GWM, GWN, NPA, false, true, 1, TERMS>);
    else if (slab) go(igemm_pp_kernel<MI, NI, GWM, GWN, NPA, false, false, 2, TERMS>);
    else go(igemm_pp_kernel<MI, NI, GWM, GWN, NPA, false, false, 1, TERMS>);
}

template <int MI, int NI, int GWM, int GWN, int NPA>
void launch_npa(const Ctx& ctx, const IGemm& p, const PPArgs& q, int items, size_t lds) {
    if (ctx.dtype == 2)
        launch_npa_terms<MI, NI, GWM, GWN, NPA, 1>(ctx, p, q, items, lds);
    else
        launch_npa_terms<MI, NI, GWM, GWN, NPA, 3>(ctx, p, q, items, lds);
}

template <int MI, int NI, int GWM, int GWN>
void launch_one(const Ctx& ctx, const IGemm& p, int Nb, const PPPlan& pl, float* part) {
    constexpr int BN = GWN * NI * 32;
    const int ncols = p.N;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (ncols + BN - 1) / BN;
    const PPGeom g = pp_geom(p);
    PPArgs q;
    q.W = p.Win;
    q.H = p.Hin;
    q.T = g.T;
    q.KW = g.KW;
    q.rowstep = g.rowstep;
    q.colstep = g.colstep;
    q.dh = p.dh;
    q.dw = p.dw;
    q.ph = p.ph;
    q.pw = p.pw;
    q.padflat = g.padflat;
    q.NLp = g.NLp;
    q.CAPl = pp_ring_lines(g, BN);
    q.ntiles = ntiles;
    q.tiles = mtiles * ntiles;
    q.nci = p.C1 / BK;
    q.cps = (q.nci + pl.S - 1) / pl.S;
    q.Nb = Nb;
    q.part = pl.S > 1 ? part : nullptr;
    q.S = pl.S;
    // slice-major items (rounds 3 - 4, again since round 6): an XCD's contiguous eighth of the items is part of ONE K slice, so its
    // L2 streams 1 / S of the packed weights; tile-major (round 5's default) gave it a few tiles with ALL their slices and every L2
    // streamed the whole weight tensor: FETCH_SIZE x 3.4 at the 5 x 39 level for the same run time (profiles/r5/r5_bf16x3_pmc_fetch_write.txt)
    q.tile_major = ctx.tune.pp_tile_major && pl.S > 1 ? 1 : 0;
    q.dbg = 0;
    q.tiles_pp = q.tiles;
    q.b_phase = 0;
    MAA_CHECK((q.nci + q.cps - 1) / q.cps == pl.S, "igemm_pp: K split leaves an empty slice");
    MAA_CHECK(q.CAPl > 0, "igemm_pp: the A ring does not fit beside the weight ring");
    const size_t lds = (size_t)q.CAPl * 128 + (size_t)NSB * BN * 128 + 128 + 1024;
    MAA_CHECK(lds <= 163840, "igemm_pp: LDS per workgroup");
    q.items = q.tiles * pl.S;
    const int cus = device_cu_count(ctx.device);
    // persistent: one workgroup per CU holds the LDS.  (Round 6, profiles/r6_call4_grid_cap_ab.txt: capping the grid at half the
    // items -- two items per persistent workgroup, the second's first copies under the first's stores, the other CUs left to
    // the other contexts' kernels -- is 1 % slower with three batches in flight and 9 % slower with one.)
    const int grid = q.items < cus ? q.items : cus;
    if (g.npa == 1)
        launch_npa<MI, NI, GWM, GWN, 1>(ctx, p, q, grid, lds);
    else
        launch_npa<MI, NI, GWM, GWN, 3>(ctx, p, q, grid, lds);
    if (pl.S > 1) launch_splitk_reduce(ctx, p, part, pl.S, q.tiles, ntiles, Nb, BM, BN, GWN, MI, NI, 512);
}

// Upsample + 3x3 convolution as four 2x2 phase convolutions in one launch (see the kernel's UP2 note).  `p` describes ONE phase:
// KH = KW = 2, the low-resolution geometry (Hin = Hout, Win = Wout), M = B H W, K = 4 C1, c = the phase-major plane buffer
// [4][M][ldc]; b_phase = floats between the phases' packed weights.
template <int MI, int NI, int GWM, int GWN>
void launch_up2(const Ctx& ctx, const IGemm& p, int Nb, long long b_phase) {
    constexpr int BN = GWN * NI * 32;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
    const PPGeom g = pp_geom(p);
    PPArgs q;
    q.W = p.Win;
    q.H = p.Hin;
    q.T = g.T;
    q.KW = g.KW;
    q.rowstep = g.rowstep;
    q.colstep = g.colstep;
    q.dh = p.dh;
    q.dw = p.dw;
    q.ph = 1;
    q.pw = 1;
    q.padflat = p.Win + 1;
    q.NLp = g.NLp;
    q.CAPl = pp_ring_lines(g, BN);
    q.ntiles = ntiles;
    q.tiles_pp = mtiles * ntiles;
    q.tiles = 4 * q.tiles_pp;
    q.nci = p.C1 / BK;
    q.cps = q.nci;
    q.Nb = Nb;
    q.part = nullptr;
    q.S = 1;
    q.tile_major = 0;
    q.b_phase = b_phase;
    q.dbg = 0;
    MAA_CHECK(q.CAPl > 0, "igemm_pp up2: the A ring does not fit beside the weight ring");
    const size_t lds = (size_t)q.CAPl * 128 + (size_t)NSB * BN * 128 + 128 + 1024;
    MAA_CHECK(lds <= 163840, "igemm_pp up2: LDS per workgroup");
    q.items = q.tiles;
    const int cus = device_cu_count(ctx.device);
    const int grid = q.items < cus ? q.items : cus;
    auto go = [&](auto kern) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, 163840);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, ctx.stream, p, q);
    };
    const bool one = ctx.dtype == 2;
    if (g.npa == 1) {
        if (one) go(igemm_pp_kernel<MI, NI, GWM, GWN, 1, false, true, 1, 1, true>);
        else go(igemm_pp_kernel<MI, NI, GWM, GWN, 1, false, true, 1, 3, true>);
    } else {
        if (one) go(igemm_pp_kernel<MI, NI, GWM, GWN, 3, false, true, 1, 1, true>);
        else go(igemm_pp_kernel<MI, NI, GWM, GWN, 3, false, true, 1, 3, true>);
    }
}


// ------------------------------------------------------------------------------------------ 1x1 / Linear
// The same two-group schedule for plain row-major contractions (q/k/v, to_out, proj_in/out, the GEGLU projection, ff.net.2):
// no halo -- a chunk's A tile is its 256 rows' lines -- so A and the weights share one 3-slot ring [256 + BN lines] and a
// wave issues NP1 = 4 + NPB pieces per memory phase.  Products are issued per accumulator in the order of the other engines
// (k ascending; lo.hi, hi.lo, hi.hi per k-step): without a K split the result is bit-identical to theirs, with S slices to
// the second engine's S-slice result.
struct PP1Args {
    int ntiles, tiles;   // N tiles, M tiles x N tiles
    int nchunks, cps;    // 32-deep chunks in all, per K slice
    int Nb;
    float* part;
};

template <int MI, int NI, int GWM, int GWN, int TERMS>
__global__ __launch_bounds__(512) void igemm_pp1_kernel(const IGemm p, const PP1Args q) {
    constexpr int BN = GWN * NI * 32;
    constexpr int BPG = BN / 16;                    // weight pieces per group and chunk
    constexpr int NPB = (BPG + 3) / 4;
    constexpr int NPA = 4;                          // A pieces per wave and memory phase: 16 per group
    constexpr int NP = NPA + NPB;
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(GWM * GWN == 4 && GWM * MI * 32 == 128, "group geometry");
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [3][A 256 lines | B BN lines][dump 1 KB]
    char* const sD = smem + NSB * STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wq = wid & 3;
    const int wm = wq / GWN, wn = wq - wm * GWN;
    const int lrow = lane & 31, lk = lane >> 5;

    int item;
    {
        const int items = (int)gridDim.x, xcd = blockIdx.x & 7, qq = items >> 3, rr = items & 7;
        item = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (int)(blockIdx.x >> 3);
    }
    const int slice = __builtin_amdgcn_readfirstlane(item / q.tiles), tile = item - slice * q.tiles;
    const int mt = __builtin_amdgcn_readfirstlane(tile / q.ntiles), nt = tile - mt * q.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int c_begin = slice * q.cps;
    const int NQ = min(q.nchunks, c_begin + q.cps) - c_begin;
    const char* const zero = reinterpret_cast<const char*>(p.zeros);

    // fragment offsets inside a stage (tile offsets are multiples of 32 rows: the swizzle depends on lrow only)
    const int swz = (lrow >> 1) & 7;
    const int a_row = (grp * 128 + wm * (32 * MI) + lrow) * 128, b_row = (BM + wn * (32 * NI) + lrow) * 128;
    int s_off[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) s_off[pl][ks] = ((pl * 4 + ks * 2 + lk) ^ swz) << 4;

    // copies: piece = 8 rows x 128 B; this wave's pieces of a chunk are A rows 8 (grp 16 + k 4 + wq) .. and weight rows
    // 8 (grp BPG + k 4 + wq) ..; running per-lane source pointers, one 128-byte step per chunk
    const int r8 = lane >> 3, sl = lane & 7;
    const char* gpa[NPA];
    const char* gpb[NPB];
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
        const int rl = 8 * (grp * 16 + k * 4 + wq) + r8;
        const long long m = min((long long)m0 + rl, (long long)p.M - 1);      // rows past the last one: clamped, never stored
        gpa[k] = reinterpret_cast<const char*>(p.a1) + (m * p.lda1 + (long long)c_begin * BK) * 4 + ((sl ^ ((rl >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
        const int nl = 8 * (grp * BPG + k * 4 + wq) + r8;
        const int n = min(n0 + nl, q.Nb - 1);
        gpb[k] = reinterpret_cast<const char*>(p.b) + ((long long)n * p.ldb + (long long)c_begin * BK) * 4 + ((sl ^ ((nl >> 1) & 7)) << 4);
    }
    // this wave's pieces of the chunk `ahead` chunks after the one its pointers stand at (ahead is 0 in steady state)
    auto issue = [&](int slot, bool live) __attribute__((always_inline)) {
        char* const st = smem + slot * STAGE;
        static_for<0, NPA>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const char* src = live ? gpa[k] : zero;
            char* dst = live ? st + (grp * 16 + k * 4 + wq) * 1024 : sD;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
            gpa[k] += 128;
        });
        static_for<0, NPB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const bool real = live && (k * 4 + wq < BPG);       // wave-uniform
            const char* src = real ? gpb[k] : zero;
            char* dst = real ? st + BM * 128 + (grp * BPG + k * 4 + wq) * 1024 : sD;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
            gpb[k] += 128;
        });
    };

    // prologue: chunks 0 and 1 (each wave its own pieces of both), then everything landed
    issue(0, true);
    issue(1, NQ > 1);
    wait_vmcnt<0>();
    __syncthreads();

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];

    auto load_phase = [&](int j, int slot) __attribute__((always_inline)) {
        const char* const st = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah[ks][i] = *reinterpret_cast<const bf16x8*>(st + a_row + i * 4096 + s_off[0][ks]);
                if constexpr (TERMS == 3) al[ks][i] = *reinterpret_cast<const bf16x8*>(st + a_row + i * 4096 + s_off[1][ks]);
            }
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bh[ks][jn] = *reinterpret_cast<const bf16x8*>(st + b_row + jn * 4096 + s_off[0][ks]);
                if constexpr (TERMS == 3) bl[ks][jn] = *reinterpret_cast<const bf16x8*>(st + b_row + jn * 4096 + s_off[1][ks]);
            }
        int slot2 = slot + (NSB - 1);
        if (slot2 >= NSB) slot2 -= NSB;
        issue(slot2, j + (NSB - 1) < NQ);       // chunk j + 2 -> the slot chunk j - 1 has left
        wait_lgkm0();
        wait_vmcnt<NP>();
    };
    auto mma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (TERMS == 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][jn], acc[i][jn], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][jn], acc[i][jn], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][jn], acc[i][jn], 0, 0, 0);
        }
    };

    if (grp == 1) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    int slot = 0;
    for (int j = 0; j < NQ; ++j) {
        load_phase(j, slot);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma_phase();
        __builtin_amdgcn_sched_barrier(0);
        if (!(grp == 1 && j == NQ - 1)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (++slot == NSB) slot = 0;
    }
    wait_vmcnt<0>();

    const int rpb = p.Hout * p.Wout;
    const int row_base = m0 + grp * 128 + wm * (32 * MI), col_base = n0 + wn * (32 * NI);
    if (q.part == nullptr) {
        igemm_epilogue<MI, NI>(p, acc, row_base, col_base, lrow, lk, 0, q.Nb, rpb);
    } else {
        float* pp = q.part + ((long long)item * (MI * NI * 4) * 512 + tid) * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jn = 0; jn < NI; ++jn)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 v = {acc[i][jn][4 * qd], acc[i][jn][4 * qd + 1], acc[i][jn][4 * qd + 2], acc[i][jn][4 * qd + 3]};
                    *reinterpret_cast<f32x4*>(pp + (long long)((i * NI + jn) * 4 + qd) * 512 * 4) = v;
                }
    }
}

template <int MI, int NI, int GWM, int GWN>
void launch_one1(const Ctx& ctx, const IGemm& p, int Nb, const PPPlan& pl, float* part) {
    constexpr int BN = GWN * NI * 32;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (ncols + BN - 1) / BN;
    PP1Args q;
    q.ntiles = ntiles;
    q.tiles = mtiles * ntiles;
    q.nchunks = p.K / BK;
    q.cps = (q.nchunks + pl.S - 1) / pl.S;
    q.Nb = Nb;
    q.part = pl.S > 1 ? part : nullptr;
    MAA_CHECK((q.nchunks + q.cps - 1) / q.cps == pl.S, "igemm_pp: K split leaves an empty slice");
    MAA_CHECK(!p.geglu || NI % 2 == 0, "GEGLU needs value / gate block pairs inside a wave");
    constexpr size_t lds = (size_t)NSB * (BM + BN) * 128 + 1024;
    static_assert(lds <= 163840, "LDS per workgroup");
    const int items = q.tiles * pl.S;
    auto go = [&](auto kern) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(512), lds, ctx.stream, p, q);
    };
    if (ctx.dtype == 2)
        go(igemm_pp1_kernel<MI, NI, GWM, GWN, 1>);
    else
        go(igemm_pp1_kernel<MI, NI, GWM, GWN, 3>);
    if (pl.S > 1) launch_splitk_reduce(ctx, p, part, pl.S, q.tiles, ntiles, Nb, BM, BN, GWN, MI, NI, 512);
}
}  // namespace

// Which problems take this engine: stride-1 "same"-padded convolutions -- the UNet's 3x3 and the vocoders' dilated 1-D kernels
// of 3 / 7 / 11 taps -- on one split32 source with split32 weights, a whole number of 32-channel chunks, and an A ring + the
// weight ring that fit the CU's LDS.  The tile width and the number of K slices depend on the layer only (never on M).
// MAA_PP = "off" | "bn,S" overrides (tuning and tests; parsed when the context is created).
PPPlan igemm_pp_plan(const Ctx& ctx, const IGemm& p) {
    PPPlan pl;
    const int T = p.KH * p.KW;
    if (!(T >= 3 && T <= 32 && p.sh == 1 && p.sw == 1 && p.up == 0)) return pl;
    if (!(2 * p.ph == p.dh * (p.KH - 1) && 2 * p.pw == p.dw * (p.KW - 1))) return pl;          // "same" padding, odd kernels
    if (!(p.KH == 1 || p.dh * p.Win >= (p.KW - 1) * p.dw)) return pl;                         // tap offsets ascend with t
    if (!(p.a_split && p.b_split && p.b_nk && p.C2 == 0 && p.C1 % BK == 0 && p.Z == 1 && p.a_act == 0 && !p.geglu)) return pl;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.K != T * p.C1 || p.N < 64) return pl;
    const PPGeom g = pp_geom(p);
    if (g.npa == 0) return pl;
    auto fits = [&](int bn) { return pp_ring_lines(g, bn) > 0; };
    const int nci = p.C1 / BK;
    int bn = 0, S = 0;
    if (!ctx.tune.pp.empty()) {
        if (ctx.tune.pp[0] == 'o') return pl;
        std::sscanf(ctx.tune.pp.c_str(), "%d,%d", &bn, &S);
    }
    if (bn != 128 && bn != 160) {
        // 160-wide tiles where they divide N (320, 640, 960, 1280): no padded columns at N = 320, and at N = 640 four K slices
        // of 4 x 13 tiles make one round of 208 workgroups (profiles/r3_pp_bench_v4_ring.txt); else 128
        bn = (p.N % 160 == 0 && fits(160)) ? 160 : 128;
    }
    if (!fits(bn)) {
        if (bn == 160 && fits(128))
            bn = 128;
        else
            return pl;
    }
    if (S <= 0) {
        if (p.KH == 1) {
            S = 1;      // the vocoders' layers: tens of thousands of rows, thousands of tiles
        } else {
            // enough (slice, tile) items for one round of ~200 workgroups at the UNet's two resolutions without the slab round
            // trip outgrowing the contraction
            const int ntiles = (p.N + bn - 1) / bn;
            S = ntiles >= 4 ? ctx.tune.pp_s_wide : ctx.tune.pp_s_narrow;
        }
    }
    if (S > nci) S = nci;
    for (; S > 1; --S) {
        const int cps = (nci + S - 1) / S;
        if ((nci + cps - 1) / cps == S) break;
    }
    pl.bn = bn;
    pl.S = S < 1 ? 1 : S;
    return pl;
}

size_t igemm_pp_workspace_floats(const IGemm& p, const PPPlan& pl) {
    if (pl.bn == 0 || pl.S <= 1) return 0;
    const long long tiles = (long long)((p.M + BM - 1) / BM) * ((p.N + pl.bn - 1) / pl.bn);
    return (size_t)(tiles * pl.S * BM * pl.bn);
}

const char* igemm_pp_name(const PPPlan& pl, int terms) {
    if (terms == 1) {
        if (pl.bn == 160) return pl.S > 1 ? "igemm_pp_bf16<256x160,splitK>" : "igemm_pp_bf16<256x160>";
        return pl.S > 1 ? "igemm_pp_bf16<256x128,splitK>" : "igemm_pp_bf16<256x128>";
    }
    if (pl.bn == 160) return pl.S > 1 ? "igemm_pp_bf16x3<256x160,splitK>" : "igemm_pp_bf16x3<256x160>";
    return pl.S > 1 ? "igemm_pp_bf16x3<256x128,splitK>" : "igemm_pp_bf16x3<256x128>";
}

void launch_igemm_pp(const Ctx& ctx, const IGemm& p, int Nb, const PPPlan& pl, float* part) {
    MAA_CHECK(pl.bn == 128 || pl.bn == 160, "igemm_pp: problem not planned for this engine");
    MAA_CHECK(pl.S == 1 || part != nullptr, "igemm_pp: split-K needs its slab workspace");
    if (pl.bn == 128)
        launch_one<2, 2, 2, 2>(ctx, p, Nb, pl, part);
    else
        launch_one<1, 5, 4, 1>(ctx, p, Nb, pl, part);
}


// The phase form of "nearest-2x upsample + 3x3 convolution" (bf16 modes, split32 source and weights, a whole number of 32-channel
// chunks, rings that fit): false = not taken, the caller runs the 3x3 convolution through the virtual upsample gather.
bool launch_igemm_pp_up2(const Ctx& ctx, const IGemm& p, int Nb, long long b_phase) {
    if (ctx.dtype == 0 || (!ctx.tune.pp.empty() && ctx.tune.pp[0] == 'o')) return false;
    if (!(p.KH == 2 && p.KW == 2 && p.sh == 1 && p.sw == 1 && p.dh == 1 && p.dw == 1 && p.up == 0)) return false;
    if (!(p.a_split && p.b_split && p.b_nk && p.C2 == 0 && p.C1 % BK == 0 && p.Z == 1 && p.a_act == 0 && !p.geglu)) return false;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.K != 4 * p.C1 || p.N < 64 || p.C1 / BK < 1) return false;
    const PPGeom g = pp_geom(p);
    if (g.npa == 0) return false;
    int bn = (p.N % 160 == 0 && pp_ring_lines(g, 160) > 0) ? 160 : 128;
    if (pp_ring_lines(g, bn) <= 0) return false;
    if (ctx.ws.dry) return true;
    const int terms = ctx.dtype == 2 ? 1 : 3;
    const char* name = terms == 1 ? (bn == 160 ? "igemm_pp_up2_bf16<256x160>" : "igemm_pp_up2_bf16<256x128>")
                                  : (bn == 160 ? "igemm_pp_up2_bf16x3<256x160>" : "igemm_pp_up2_bf16x3<256x128>");
    ProfScope prof(ctx, name, 2.0 * 4.0 * p.M * (double)p.N * p.K, 4.0 * (4.0 * p.K * p.N + 4.0 * p.M * p.N));
    if (bn == 128)
        launch_up2<2, 2, 2, 2>(ctx, p, Nb, b_phase);
    else
        launch_up2<1, 5, 4, 1>(ctx, p, Nb, b_phase);
    MAA_HIP(hipGetLastError());
    return true;
}

// 1x1 / Linear problems (both operands split32, one source, a whole number of 32-deep chunks).  Without a K split the engine
// is bit-identical to the others, so taking it may depend on M: only when the 256-row tiles fill a useful part of the chip.
// The number of K slices follows the second engine's layer-only rule (K >= 2048: two slices).  MAA_PP1 = "off" | "bn,S".
PPPlan igemm_pp1_plan(const Ctx& ctx, const IGemm& p) {
    PPPlan pl;
    if (!(p.KH == 1 && p.KW == 1 && p.a_split && p.b_split && p.b_nk && p.C2 == 0 && p.Z == 1 && p.a_act == 0 && p.up == 0)) return pl;
    if (p.K % BK != 0 || p.K != p.C1 || p.K < 64) return pl;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    int bn = 0, S = 0;
    if (!ctx.tune.pp1.empty()) {
        if (ctx.tune.pp1[0] == 'o') return pl;
        std::sscanf(ctx.tune.pp1.c_str(), "%d,%d", &bn, &S);
    }
    const bool forced = bn == 128 || bn == 160;
    if (!forced) {
        bn = (!p.geglu && ncols % 160 == 0 && ncols % 128 != 0) ? 160 : 128;
        // Measured (profiles/r3_pp_bench_v2.txt): with 256-row tiles these short contractions are bound by how evenly the
        // tiles fill 256 CUs, not by the K loop -- the form only wins where the tiles make several nearly full rounds (the
        // GEGLU projection at 10x78: 980 tiles = 3.8 rounds, -7 %); everywhere else the smaller-tile engines stay.
        const long long tiles = (long long)((p.M + BM - 1) / BM) * ((ncols + bn - 1) / bn);
        const long long rounds = (tiles + 255) / 256;
        if (ncols < 128 || tiles < 768 || rounds * 256 - tiles > 64) return pl;
    }
    if (p.geglu) bn = 128;
    const int nchunks = p.K / BK;
    if (S <= 0) S = (p.K >= 2048 && !p.geglu) ? 2 : 1;
    if (S > nchunks) S = nchunks;
    for (; S > 1; --S) {
        const int cps = (nchunks + S - 1) / S;
        if ((nchunks + cps - 1) / cps == S) break;
    }
    pl.bn = bn;
    pl.S = S < 1 ? 1 : S;
    return pl;
}

size_t igemm_pp1_workspace_floats(const IGemm& p, const PPPlan& pl) {
    if (pl.bn == 0 || pl.S <= 1) return 0;
    const int ncols = p.N * (p.geglu ? 2 : 1);
    const long long tiles = (long long)((p.M + BM - 1) / BM) * ((ncols + pl.bn - 1) / pl.bn);
    return (size_t)(tiles * pl.S * BM * pl.bn);
}

const char* igemm_pp1_name(const PPPlan& pl, int terms) {
    if (terms == 1) {
        if (pl.bn == 160) return pl.S > 1 ? "igemm_pp1_bf16<256x160,splitK>" : "igemm_pp1_bf16<256x160>";
        return pl.S > 1 ? "igemm_pp1_bf16<256x128,splitK>" : "igemm_pp1_bf16<256x128>";
    }
    if (pl.bn == 160) return pl.S > 1 ? "igemm_pp1_bf16x3<256x160,splitK>" : "igemm_pp1_bf16x3<256x160>";
    return pl.S > 1 ? "igemm_pp1_bf16x3<256x128,splitK>" : "igemm_pp1_bf16x3<256x128>";
}

void launch_igemm_pp1(const Ctx& ctx, const IGemm& p, int Nb, const PPPlan& pl, float* part) {
    MAA_CHECK(pl.bn == 128 || pl.bn == 160, "igemm_pp1: problem not planned for this engine");
    MAA_CHECK(pl.S == 1 || part != nullptr, "igemm_pp1: split-K needs its slab workspace");
    if (pl.bn == 128)
        launch_one1<2, 2, 2, 2>(ctx, p, Nb, pl, part);
    else
        launch_one1<1, 5, 4, 1>(ctx, p, Nb, pl, part);
}

}  // namespace maa
