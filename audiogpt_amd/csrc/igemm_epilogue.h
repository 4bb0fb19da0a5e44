// Shared epilogue of the implicit-GEMM kernels.  The C/D fragment map of the 32x32 MFMA shapes is the same
// for every input dtype on gfx950 (lane l, register r: column l&31, row (r&3) + 8*(r>>2) + 4*(l>>5)), so the
// fp32 and the bf16-split kernels share this code.
#pragma once
#include "maa_internal.h"

namespace maa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
                                const float gl = 0.5f * g * (1.f + erff(g * 0.70710678118654752440f));
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
                        v *= p.out_scale;
                        float* dst = cp + (long long)m * p.ldc + n;
                        if (p.accumulate) v += *dst;
                        *dst = v;
                    }
                }
            }
        }
}

}  // namespace maa
