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

// one output element in the split32 form (row pitch unchanged: every 32 columns = [32 bf16 hi | 32 bf16 lo])
__device__ __forceinline__ void store_split1(float* row, int n, float v) {
    unsigned short* o = reinterpret_cast<unsigned short*>(row) + (n >> 5) * 64 + (n & 31);
    const __bf16 h = (__bf16)v;                                   // v_cvt_pk_bf16_f32, round to nearest even
    const unsigned short hb = __builtin_bit_cast(unsigned short, h);
    const __bf16 l = (__bf16)(v - __builtin_bit_cast(float, (unsigned)hb << 16));
    o[0] = hb;
    o[32] = __builtin_bit_cast(unsigned short, l);
}

// acc[MI][NI]: MI x NI fragments of 32x32 owned by this wave; (m_base, n_base) = first row / column of the wave.
template <int MI, int NI>
__device__ __forceinline__ void igemm_epilogue(const IGemm& p, f32x16 (&acc)[MI][NI], int m_base, int n_base,
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
                        const float bv = p.bias ? p.bias[cpk] : 0.f;
                        const float bg = p.bias ? p.bias[cpk + 32] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                            if (m < p.M) {
                                const float val = acc[i][j][r] * p.alpha + bv;
                                const float g = acc[i][j + 1][r] * p.alpha + bg;
                                const float gl = 0.5f * g * (1.f + fast_erff(g * 0.70710678118654752440f));
                                if (p.c_split)
                                    store_split1(cp + (long long)m * p.ldc, ncol, val * gl);
                                else
                                    cp[(long long)m * p.ldc + ncol] = val * gl;
                            }
                        }
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n_base + j * 32 + lrow;
            if (n < p.N) {
                const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (m < p.M) {
                        float v = acc[i][j][r] * p.alpha + bias;
                        if (p.rowadd) v += p.rowadd[(long long)(m / rpb) * p.ld_rowadd + n];
                        if (resp) v += resp[(long long)m * p.ldr + n];
                        if (p.act == 1) v = tanhf(v);
                        else if (p.act == 2) v = fmaxf(v, 0.f);
                        v *= p.out_scale;
                        if (p.c_split) {
                            store_split1(cp + (long long)m * p.ldc, n, v);
                        } else {
                            float* dst = cp + (long long)m * p.ldc + n;
                            if (p.accumulate) v += *dst;
                            *dst = v;
                        }
                    }
                }
            }
        }
}

}  // namespace maa
