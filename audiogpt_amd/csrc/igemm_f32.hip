// Implicit-GEMM engine, exact-fp32 path: v_mfma_f32_32x32x2_f32 (gfx950), LDS-staged tiles.
//
// One kernel covers every dense contraction on the Make-An-Audio hot path:
//   conv2d 3x3 s1/s2 (+ virtual nearest-2x upsample, + virtual channel concat of two sources)
//       reference: openaimodel.py:204,230,151-153,116-118,738  model.py:47-57,93-108
//   linear / conv1x1 (taps = 1)         attention.py:161-168,40,60,233-248  openaimodel.py:241,304,312
//   dilated conv1d k in {3,7,11}        NeuralSeq/modules/hifigan/hifigan.py:34-51 (sequence = image with H=1)
//   ConvTranspose1d polyphase groups    hifigan.py:121-125 (see pack.cpp)
//   batched Q.K^T and P.V               attention.py:178-192, openaimodel.py:366-371, model.py:186-198
// Fused: activation while staging A (leaky-ReLU / SiLU), bias, per-sample row add (time embedding,
// openaimodel.py:264-273), residual add (:275), GEGLU (attention.py:42-44), tanh, MRF accumulate
// (hifigan.py:158-164).
//
// Layout: A rows are output positions (channels-last activations), K runs (ky, kx, ci); B is the
// packed weight [K][N].  LDS tiles are k-major (A_lds[k][m], B_lds[k][n]) so that the f32 MFMA
// operands (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) are two conflict-free ds_read_b32 rows.
// Global->register->LDS double buffering, one barrier per 16-deep K chunk.
#include "maa_internal.h"

namespace maa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int NT = 256;

__device__ __forceinline__ float apply_a_act(float v, int act, float slope) {
    if (act == 1) return v > 0.f ? v : v * slope;
    if (act == 2) return v / (1.f + expf(-v));
    return v;
}

template <int BM, int BN, int WGM, int WGN, bool B_NK>
__global__ __launch_bounds__(NT) void igemm_f32_kernel(const IGemm p, int ntiles, int fast_a, int Nb) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int LDA = BM + 2;
    constexpr int LDB = B_NK ? BN + 2 : BN;
    constexpr int AL = BM / 64;                       // float4 of A per thread per chunk
    constexpr int BL = (BN * 4 + NT - 1) / NT;        // float4 of B per thread per chunk
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(BM % 64 == 0, "BM");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int nt = bid % ntiles, mt = bid / ntiles;
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

    // ---- per-thread A row bookkeeping
    const int kq = tid & 3;
    int a_b[AL], a_iy0[AL], a_ix0[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const int m = m0 + (tid >> 2) + 64 * j;
        if (m < p.M) {
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

    float4 ra[AL];
    float4 rb[BL];

    auto load_a = [&](int k0, int tap, int ci0) {
        if (fast_a) {
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int kk = ci0 + kq * 4;
#pragma unroll
            for (int j = 0; j < AL; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                int iy = a_iy0[j] + ky * p.dh, ix = a_ix0[j] + kx * p.dw;
                const bool ok = a_b[j] >= 0 && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim;
                if (ok) {
                    iy >>= p.up;
                    ix >>= p.up;
                    const long long off = ((long long)a_b[j] * p.Hin + iy) * p.Win + ix;
                    const float* src;
                    int cend;
                    if (kk < p.C1) {
                        src = a1 + off * p.lda1 + kk;
                        cend = p.C1;
                    } else {
                        src = a2 + off * p.lda2 + (kk - p.C1);
                        cend = Ctot;
                    }
                    if (kk + 3 < cend) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (kk + 0 < cend) v.x = src[0];
                        if (kk + 1 < cend) v.y = src[1];
                        if (kk + 2 < cend) v.z = src[2];
                    }
                    if (p.a_act) {
                        v.x = apply_a_act(v.x, p.a_act, p.a_slope);
                        v.y = apply_a_act(v.y, p.a_act, p.a_slope);
                        v.z = apply_a_act(v.z, p.a_act, p.a_slope);
                        v.w = apply_a_act(v.w, p.a_act, p.a_slope);
                    }
                }
                ra[j] = v;
            }
        } else {
            // generic gather: any channel count / alignment (first convs with Cin = 1, 4, 9)
#pragma unroll
            for (int j = 0; j < AL; ++j) {
                float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kg = k0 + kq * 4 + q;
                    if (kg < p.K && a_b[j] >= 0) {
                        const int tp = kg / Ctot, ci = kg - tp * Ctot;
                        const int ky = tp / p.KW, kx = tp - ky * p.KW;
                        int iy = a_iy0[j] + ky * p.dh, ix = a_ix0[j] + kx * p.dw;
                        if (iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim) {
                            iy >>= p.up;
                            ix >>= p.up;
                            const long long off = ((long long)a_b[j] * p.Hin + iy) * p.Win + ix;
                            const float x = ci < p.C1 ? a1[off * p.lda1 + ci] : a2[off * p.lda2 + (ci - p.C1)];
                            e[q] = apply_a_act(x, p.a_act, p.a_slope);
                        }
                    }
                }
                ra[j] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };

    auto load_b = [&](int k0) {
        if (B_NK) {
#pragma unroll
            for (int j = 0; j < BL; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int nrow = (tid >> 2) + 64 * j;
                const int n = n0 + nrow, k = k0 + kq * 4;
                if (nrow < BN && n < p.N && k < p.K) {
                    const float* src = bp + (long long)n * p.ldb + k;
                    if (k + 3 < p.K) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        v.x = src[0];
                        if (k + 1 < p.K) v.y = src[1];
                        if (k + 2 < p.K) v.z = src[2];
                    }
                }
                rb[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < BL; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int idx = tid + NT * j;
                const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
                const int k = k0 + kr, n = n0 + nq * 4;
                if (kr < BK && k < p.K && n < Nb) v = *reinterpret_cast<const float4*>(bp + (long long)k * p.ldb + n);
                rb[j] = v;
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* A = As + buf * BK * LDA;
        float* B = Bs + buf * BK * LDB;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            const int row = (tid >> 2) + 64 * j;
            A[(kq * 4 + 0) * LDA + row] = ra[j].x;
            A[(kq * 4 + 1) * LDA + row] = ra[j].y;
            A[(kq * 4 + 2) * LDA + row] = ra[j].z;
            A[(kq * 4 + 3) * LDA + row] = ra[j].w;
        }
        if (B_NK) {
#pragma unroll
            for (int j = 0; j < BL; ++j) {
                const int nrow = (tid >> 2) + 64 * j;
                if (nrow < BN) {
                    B[(kq * 4 + 0) * LDB + nrow] = rb[j].x;
                    B[(kq * 4 + 1) * LDB + nrow] = rb[j].y;
                    B[(kq * 4 + 2) * LDB + nrow] = rb[j].z;
                    B[(kq * 4 + 3) * LDB + nrow] = rb[j].w;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < BL; ++j) {
                const int idx = tid + NT * j;
                const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
                if (kr < BK) *reinterpret_cast<float4*>(&B[kr * LDB + nq * 4]) = rb[j];
            }
        }
    };

    // ---- accumulators
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
    const int a_base = wm * WTM + lrow, b_base = wn * WTN + lrow;

    const int nchunks = (p.K + BK - 1) / BK;
    int tap = 0, ci0 = 0;
    load_a(0, tap, ci0);
    load_b(0);
    store_tiles(0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) {
            ci0 += BK;
            if (ci0 >= Ctot && p.KH * p.KW > 1) {
                ci0 = 0;
                ++tap;
            }
            load_a((c + 1) * BK, tap, ci0);
            load_b((c + 1) * BK);
        }
        const float* A = As + buf * BK * LDA;
        const float* B = Bs + buf * BK * LDB;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = A[(kk + lk) * LDA + a_base + i * 32];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = B[(kk + lk) * LDB + b_base + j * 32];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nchunks) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    float* cp = p.c + coff;
    const float* resp = p.res ? p.res + coff : nullptr;
    if (p.geglu) {
        // packed columns: [32 value | 32 gate] per group of 64; output column = group*32 + j
        if constexpr (NI % 2 == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; j += 2) {
                    const int cpk = n0 + wn * WTN + j * 32 + lrow;   // packed value column
                    const int ncol = (cpk >> 6) * 32 + lrow;         // output column
                    if (cpk + 32 < Nb && ncol < p.N) {
                        const float bv = p.bias ? p.bias[cpk] : 0.f;
                        const float bg = p.bias ? p.bias[cpk + 32] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
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
            const int n = n0 + wn * WTN + j * 32 + lrow;
            if (n < p.N) {
                const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
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

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(const Ctx& ctx, const IGemm& p, int fast_a, int Nb) {
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N * (p.geglu ? 2 : 1) + BN - 1) / BN;
    dim3 grid((unsigned)((long long)mtiles * ntiles), (unsigned)p.Z);
    if (p.b_nk)
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, true>), grid, dim3(NT), 0, ctx.stream, p, ntiles,
                           fast_a, Nb);
    else
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, false>), grid, dim3(NT), 0, ctx.stream, p, ntiles,
                           fast_a, Nb);
}

inline double tile_cost(long long M, long long N, int Z, int BM, int BN, double eff) {
    const long long blocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * Z;
    const long long rounds = (blocks + 255) / 256;
    return (double)rounds * BM * BN / eff;
}

}  // namespace

void launch_igemm(const Ctx& ctx, const IGemm& p) {
    if (ctx.ws.dry) return;
    MAA_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "empty igemm");
    MAA_CHECK(p.K == p.KH * p.KW * (p.C1 + p.C2), "igemm K mismatch");
    const int taps = p.KH * p.KW, Ctot = p.C1 + p.C2;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    int fast_a = (taps == 1 || (Ctot % BK == 0 && p.C1 % BK == 0)) && p.lda1 % 4 == 0 && al16(p.a1) &&
                 p.a_so % 4 == 0 && p.a_si % 4 == 0;
    if (p.C2 > 0) fast_a = fast_a && p.C1 % 4 == 0 && p.lda2 % 4 == 0 && al16(p.a2);
    if (p.Z > 1) MAA_CHECK(p.C2 == 0, "batched igemm takes one A source");
    // columns that may be read from B: packed weights are zero-padded to a multiple of 32
    const int ncols = p.N * (p.geglu ? 2 : 1);
    int Nb = ncols;
    if (!p.b_nk) {
        MAA_CHECK(p.ldb % 4 == 0 && al16(p.b) && p.b_so % 4 == 0 && p.b_si % 4 == 0, "B [K][N] alignment");
        Nb = (ncols + 3) / 4 * 4;
        MAA_CHECK(Nb <= p.ldb || p.Z == 1, "B column padding");
        if (Nb > p.ldb) Nb = p.ldb / 4 * 4;
    } else {
        MAA_CHECK(p.ldb % 4 == 0 && al16(p.b) && p.b_so % 4 == 0 && p.b_si % 4 == 0, "B [N][K] alignment");
    }
    // algorithmic work of this launch: 2*M*N*K per batch entry (GEGLU computes 2N columns)
    const double flops = 2.0 * p.M * (double)ncols * p.K * p.Z;
    const double bytes = 4.0 * ((double)p.K * ncols + (double)p.M * p.N * p.Z);   // weights once + output once
    int cfg = 0;
    if (p.geglu) {
        MAA_CHECK(ncols % 64 == 0, "geglu needs packed N multiple of 64");
        cfg = 0;
    } else if (ncols <= 32) {
        cfg = 3;
    } else {
        const double c128 = tile_cost(p.M, ncols, p.Z, 128, 128, 1.00);
        const double c12864 = tile_cost(p.M, ncols, p.Z, 128, 64, 0.92);
        const double c64 = tile_cost(p.M, ncols, p.Z, 64, 64, 0.80);
        cfg = (c128 <= c12864 && c128 <= c64) ? 0 : (c12864 <= c64 ? 1 : 2);
    }
    static const char* kNames[4] = {"igemm_f32<128x128>", "igemm_f32<128x64>", "igemm_f32<64x64>", "igemm_f32<256x32>"};
    ProfScope prof(ctx, kNames[cfg], flops, bytes);
    switch (cfg) {
        case 0: launch_cfg<128, 128, 2, 2>(ctx, p, fast_a, Nb); break;
        case 1: launch_cfg<128, 64, 2, 2>(ctx, p, fast_a, Nb); break;
        case 2: launch_cfg<64, 64, 2, 2>(ctx, p, fast_a, Nb); break;
        default: launch_cfg<256, 32, 4, 1>(ctx, p, fast_a, Nb); break;
    }
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
