// Shared epilogue of the implicit-GEMM kernels.  The C/D fragment map of the 32x32 MFMA shapes is the same
// for every input dtype on gfx950 (lane l, register r: column l&31, row (r&3) + 8*(r>>2) + 4*(l>>5)), so the
// fp32 and the bf16-split kernels share this code.
#pragma once
#include "maa_internal.h"

namespace maa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// erf for the GELU of the GEGLU epilogue: branch-free, ~25 VALU instructions (libm's erff costs ~100 and was a
// third of the ff.net.0 launch).  |x| < 1: x * P5(x^2); else 1 - exp(-Q7(|x|)) with |x| clamped to 4 (erf = 1 in
// fp32 beyond).  Polynomials are Chebyshev-node fits; max |error| against the fp64 erf is 1.2e-7 (0.5-1 ulp of
// the result near 1), measured over [-6, 6] in fp32 arithmetic.
__device__ __forceinline__ float fast_erff(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    const float s = t * t;
    float r = -5.654105860e-04f;
    r = fmaf(r, s, 4.923277665e-03f);
    r = fmaf(r, s, -2.671638510e-02f);
    r = fmaf(r, s, 1.128036441e-01f);
    r = fmaf(r, s, -3.761234978e-01f);
    r = fmaf(r, s, 1.128379127e+00f);
    const float small = r * t;
    float q = 1.330939039e-05f;
    q = fmaf(q, t, -3.175720346e-04f);
    q = fmaf(q, t, 3.436867598e-03f);
    q = fmaf(q, t, -2.262198592e-02f);
    q = fmaf(q, t, 1.033189800e-01f);
    q = fmaf(q, t, 6.390933195e-01f);
    q = fmaf(q, t, 1.125925949e+00f);
    q = fmaf(q, t, 7.569686250e-04f);
    const float large = 1.0f - __builtin_amdgcn_exp2f(q * -1.4426950408889634f);
    return copysignf(t < 1.0f ? small : large, x);
}

// Pins a loaded value: the load is waited for HERE, once, in straight-line code.  Without it the compiler sinks the first
// use into the per-row `if (m < M)` branches and has to wait there -- with vmcnt(0), i.e. also for every store issued by
// the earlier rows, which turns a block's 16 stores into 16 dependent round trips.
__device__ __forceinline__ void settle(float& x) { asm volatile("" : "+v"(x)); }

// An output element in the split32 form (row pitch unchanged: every 32 columns = [32 bf16 hi | 32 bf16 lo]), from the
// accumulator layout of the MFMA epilogues, where lane l holds column n = ... + (l & 31): the two lanes of an
// even / odd column pair swap one half each (DPP quad_perm [1,0,3,2]) so that the even lane stores both hi halves and the odd
// lane both lo halves -- ONE 4-byte store per lane and element instead of two 2-byte stores.  Both lanes of a pair must be
// active (n even <-> lane even; callers guarantee N % 2 == 0 and a row condition that is uniform over the pair).
__device__ __forceinline__ void store_split_pair(float* row, int n, float v) {
    const __bf16 h = (__bf16)v;
    const unsigned hb = __builtin_bit_cast(unsigned short, h);
    const __bf16 l = (__bf16)(v - __builtin_bit_cast(float, hb << 16));
    const unsigned lb = __builtin_bit_cast(unsigned short, l);
    const bool even = (n & 1) == 0;
    const unsigned mine = even ? lb : hb;                                  // what the partner stores
    const unsigned theirs = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);
    unsigned* o = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(row) + (n >> 5) * 64 + ((n & 31) & ~1) + (even ? 0 : 32));
    *o = even ? (hb | (theirs << 16)) : (theirs | (lb << 16));
}

// acc[MI][NI]: MI x NI fragments of 32x32 owned by this wave; (m_base, n_base) = first row / column of the wave.
// PAIR: plain fp32 output whose columns pair up (even width, even pitch, 8-byte aligned base): see the store loop
template <int MI, int NI, bool PAIR>
__device__ __forceinline__ void igemm_epilogue_impl(const IGemm& p, f32x16 (&acc)[MI][NI], int m_base, int n_base,
                                                    int lrow, int lk, long long coff, int Nb, int rpb) {
    float* cp = p.c + coff;
    const float* resp = p.res ? p.res + coff : nullptr;
    if (p.geglu) {
        // packed columns: [32 value | 32 gate] per group of 64; output column = group*32 + j  (attention.py:42-44)
        if constexpr (NI % 2 == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; j += 2) {
                    const int cpk = n_base + j * 32 + lrow;          // packed value column
                    const int ncol = (cpk >> 6) * 32 + lrow;         // output column
                    if (cpk + 32 < Nb && ncol < p.N) {
                        float bv = p.bias ? p.bias[cpk] : 0.f;
                        float bg = p.bias ? p.bias[cpk + 32] : 0.f;
                        settle(bv);
                        settle(bg);
                        // values first (straight-line: the bias loads are waited for once), stores after -- a load result
                        // first used inside a per-row branch makes every branch wait for all earlier stores as well
                        float outv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float val = acc[i][j][r] * p.alpha + bv;
                            const float g = acc[i][j + 1][r] * p.alpha + bg;
                            const float gl = 0.5f * g * (1.f + fast_erff(g * 0.70710678118654752440f));
                            outv[r] = val * gl;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                            if (m < p.M) {
                                if (p.c_split)
                                    store_split_pair(cp + (long long)m * p.ldc, ncol, outv[r]);
                                else
                                    cp[(long long)m * p.ldc + ncol] = outv[r];
                            }
                        }
                    }
                }
        }
        return;
    }
    // Every global read of a 32x32 block (time-embedding row add, residual, accumulate target) is issued before the block's
    // first store, so a block costs ONE memory round trip.  (Written as load-compute-store per element, the possible
    // aliasing of `res` / `c` makes the compiler wait for each load and each store in turn: 32-48 dependent round trips
    // per block -- 10-20 us per workgroup, which was the dominant cost of the short-K layers up to round 2.)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        // sample index of the block's rows for the per-sample row add: 32 consecutive rows span at most two samples when a
        // sample has >= 32 rows (one division per block instead of one per row)
        const int mb = m_base + i * 32;
        const int b0 = mb / rpb;
        const int nextb = (b0 + 1) * rpb;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n_base + j * 32 + lrow;
            if (n < p.N) {
                float bias = p.bias ? p.bias[n] : 0.f;
                float ra[16], rs[16], cv[16];
                // unconditional loads from clamped rows under wave-uniform "is this term present" branches: straight-line
                // code, all of a block's reads in flight together
                long long mc[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    mc[r] = m < p.M ? m : p.M - 1;
                    ra[r] = rs[r] = cv[r] = 0.f;
                }
                if (p.rowadd) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int bb = rpb >= 32 ? (mc[r] < nextb ? b0 : b0 + 1) : (int)(mc[r] / rpb);
                        ra[r] = p.rowadd[(long long)bb * p.ld_rowadd + n];
                    }
                }
                if (resp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rs[r] = resp[mc[r] * p.ldr + n];
                }
                if (p.accumulate && !p.c_split) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cv[r] = cp[mc[r] * p.ldc + n];
                }
                // values first (straight-line: every read above is waited for once), then the stores
                settle(bias);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    settle(ra[r]);
                    settle(rs[r]);
                    settle(cv[r]);
                }
                float outv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * p.alpha + bias;
                    if (p.rowadd) v += ra[r];
                    if (resp) v += rs[r];
                    if (p.act == 1) v = tanhf(v);
                    else if (p.act == 2) v = fmaxf(v, 0.f);
                    else if (p.act == 3) v = 0.5f * v * (1.f + fast_erff(v * 0.70710678118654752440f));
                    else if (p.act == 4) v = v > 0.f ? v : v * p.act_slope;
                    v *= p.out_scale;
                    if (p.accumulate && !p.c_split) v += cv[r];
                    outv[r] = v;
                }
                if constexpr (PAIR) {
                    // fp32 rows, two columns per lane: the lanes of an even / odd column pair swap one value per pair of
                    // rows (DPP), the even lane stores columns (n, n + 1) of the even row, the odd lane those of the odd row
                    // -- 8 eight-byte stores per lane and block instead of 16 four-byte ones (the tail of these kernels is
                    // store-issue-bound)
                    const bool even = (lrow & 1) == 0;
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp) {
                        const float keep = even ? outv[2 * rp] : outv[2 * rp + 1];
                        const float give = even ? outv[2 * rp + 1] : outv[2 * rp];
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, false));
                        const int r = 2 * rp + (even ? 0 : 1);
                        const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (m < p.M) {
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 v2 = even ? f32x2{keep, got} : f32x2{got, keep};
                            *reinterpret_cast<f32x2*>(cp + (long long)m * p.ldc + (n & ~1)) = v2;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (m < p.M) {
                            if (p.c_split)
                                store_split_pair(cp + (long long)m * p.ldc, n, outv[r]);
                            else
                                cp[(long long)m * p.ldc + n] = outv[r];
                            if (p.c2) {
                                const float w = outv[r] > 0.f ? outv[r] : outv[r] * p.c2_slope;
                                store_split_pair(p.c2 + coff + (long long)m * p.ldc2, n, w);
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int MI, int NI>
__device__ __forceinline__ void igemm_epilogue(const IGemm& p, f32x16 (&acc)[MI][NI], int m_base, int n_base, int lrow,
                                               int lk, long long coff, int Nb, int rpb) {
    const bool pair = !p.no_pair && !p.geglu && !p.c_split && p.c2 == nullptr && (p.N & 1) == 0 && (p.ldc & 1) == 0 && (coff & 1) == 0 &&
                      (reinterpret_cast<uintptr_t>(p.c) & 7) == 0;
    if (pair)
        igemm_epilogue_impl<MI, NI, true>(p, acc, m_base, n_base, lrow, lk, coff, Nb, rpb);
    else
        igemm_epilogue_impl<MI, NI, false>(p, acc, m_base, n_base, lrow, lk, coff, Nb, rpb);
}

}  // namespace maa
