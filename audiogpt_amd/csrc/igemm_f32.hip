// Implicit-GEMM engine, exact-fp32 path: v_mfma_f32_32x32x2_f32 (gfx950), LDS-staged tiles.
//
// One kernel covers every dense contraction on the Make-An-Audio hot path:
//   conv2d 3x3 s1/s2 (+ virtual nearest-2x upsample, + virtual channel concat of two sources)
//       reference: openaimodel.py:204,230,151-153,116-118,738  model.py:47-57,93-108
//   linear / conv1x1 (taps = 1)         attention.py:161-168,40,60,233-248  openaimodel.py:241,304,312
//   dilated conv1d k in {3,7,11}        NeuralSeq/modules/hifigan/hifigan.py:34-51 (sequence = image with H=1)
//   ConvTranspose1d polyphase groups    hifigan.py:121-125 (see runtime.cpp)
//   batched Q.K^T and P.V               attention.py:178-192, openaimodel.py:366-371, model.py:186-198
// Fused: leaky-ReLU while staging A, bias, per-sample row add (time embedding, openaimodel.py:264-273),
// residual add (:275), GEGLU (attention.py:42-44), tanh, MRF accumulate (hifigan.py:158-164).
//
// Layout: A rows are output positions (channels-last activations), K runs (ky, kx, ci); B is the packed
// weight [K][N].  LDS tiles are k-major (A_lds[k][m], B_lds[k][n]) so that the f32 MFMA operands
// (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) are two conflict-free ds_read_b32 rows.
// Pipeline: global -> registers two K-chunks ahead (two register sets) -> LDS double buffer, one barrier per
// 16-deep chunk.  The gather state (row pointers, validity) is recomputed only when the tap changes, so the
// steady-state loop is loads + LDS traffic + MFMA with no integer division or 64-bit multiplies.
#include "maa_internal.h"

namespace maa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int NT = 256;

// GENERIC = true: per-element gather for convs whose channel count is not a multiple of 16 (first convs with
// Cin = 1, 4, 9, 80) -- tiny share of the work, kept out of the fast kernel's loop.
template <int BM, int BN, int WGM, int WGN, bool B_NK, bool GENERIC>
__global__ __launch_bounds__(NT) void igemm_f32_kernel(const IGemm p, int ntiles, int Nb) {
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
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own 4 MB L2), so consecutive TILES are
    // handed to the SAME XCD -- the N-tiles of one M-tile and the neighbouring M-tiles (which share A rows through
    // the conv halo) then hit in one L2 instead of being fetched over the fabric once per XCD.  Bijective for any
    // grid size; a pure speed choice, results do not depend on placement.
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, xcd = bid & 7, qq = nblk >> 3, rr = nblk & 7;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
    }
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
    const int taps = p.KH * p.KW;
    // Masked-out tile elements are loaded from a zero page instead of being branched around: every global load
    // in the K loop is then unconditional, so the prefetches stay in flight behind counted s_waitcnt vmcnt(N).
    const float4* g_zero4 = reinterpret_cast<const float4*>(p.zeros);
    const float slope = p.a_act == 1 ? p.a_slope : 1.0f;     // leaky(x) = max(x, slope*x); slope 1 = identity

    // ---- per-thread A rows: output position -> (batch, top-left input coordinate)
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

    // ---- gather state of the tap being streamed
    const float* a_p1[AL];
    const float* a_p2[AL];
    bool a_ok[AL];
    int g_tap = 0, g_ci = 0;      // next chunk to load: tap index, first channel
    auto set_tap = [&](int tap) {
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int iy = a_iy0[j] + ky * p.dh, ix = a_ix0[j] + kx * p.dw;
            const bool ok = a_b[j] >= 0 && iy >= 0 && iy < Hlim && ix >= 0 && ix < Wlim;
            iy >>= p.up;
            ix >>= p.up;
            const long long off = ok ? ((long long)a_b[j] * p.Hin + iy) * p.Win + ix : 0;
            a_ok[j] = ok;
            a_p1[j] = a1 + off * p.lda1 + kq * 4;
            a_p2[j] = a2 + off * p.lda2 + kq * 4 - p.C1;
        }
    };

    auto load_a = [&](float4 (&ra)[AL], int k0) {
        if constexpr (!GENERIC) {
            // chunk [g_ci, g_ci+16) of tap g_tap lies inside one source (C1 % 16 == 0 or single source)
            const bool first = g_ci < p.C1;
            const int cend = first ? p.C1 : Ctot;
            const bool kin = g_ci + kq * 4 < cend;           // K tail (taps == 1, K % 16 != 0, K % 4 == 0)
#pragma unroll
            for (int j = 0; j < AL; ++j) {
                const float4* src = reinterpret_cast<const float4*>((first ? a_p1[j] : a_p2[j]) + g_ci);
                ra[j] = *((a_ok[j] && kin) ? src : g_zero4);
                // (the activation is applied when the tile is written to LDS: the loaded value is not used
                //  here, so the load stays in flight across the MFMA block)
            }
            g_ci += BK;
            if (g_ci >= Ctot && g_tap + 1 < taps) {
                g_ci = 0;
                ++g_tap;
                set_tap(g_tap);
            }
        } else {
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
                            e[q] = x;
                        }
                    }
                }
                ra[j] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };

    // ---- B rows / columns of this thread (fixed), pointer advanced per chunk
    const float* b_ptr[BL];
    bool b_ok[BL];
#pragma unroll
    for (int j = 0; j < BL; ++j) {
        if (B_NK) {
            const int nrow = (tid >> 2) + 64 * j;
            const int n = n0 + nrow;
            b_ok[j] = nrow < BN && n < Nb;
            b_ptr[j] = bp + (long long)(b_ok[j] ? n : 0) * p.ldb + kq * 4;
        } else {
            const int idx = tid + NT * j;
            const int kr = idx / (BN / 4), nq = idx - kr * (BN / 4);
            const int n = n0 + nq * 4;
            b_ok[j] = kr < BK && n < Nb;
            b_ptr[j] = bp + (long long)kr * p.ldb + (b_ok[j] ? n : 0);
        }
    }
    auto load_b = [&](float4 (&rb)[BL], int k0) {
#pragma unroll
        for (int j = 0; j < BL; ++j) {
            if (B_NK) {
                const float4* src = reinterpret_cast<const float4*>(b_ptr[j] + k0);
                rb[j] = *((b_ok[j] && k0 + kq * 4 < p.K) ? src : g_zero4);
            } else {
                const int kr = (tid + NT * j) / (BN / 4);
                const float4* src = reinterpret_cast<const float4*>(b_ptr[j] + (long long)k0 * p.ldb);
                rb[j] = *((b_ok[j] && k0 + kr < p.K) ? src : g_zero4);
            }
        }
    };

    auto store_tiles = [&](const float4 (&ra)[AL], const float4 (&rb)[BL], int buf) {
        float* A = As + buf * BK * LDA;
        float* B = Bs + buf * BK * LDB;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            const int row = (tid >> 2) + 64 * j;
            A[(kq * 4 + 0) * LDA + row] = fmaxf(ra[j].x, ra[j].x * slope);
            A[(kq * 4 + 1) * LDA + row] = fmaxf(ra[j].y, ra[j].y * slope);
            A[(kq * 4 + 2) * LDA + row] = fmaxf(ra[j].z, ra[j].z * slope);
            A[(kq * 4 + 3) * LDA + row] = fmaxf(ra[j].w, ra[j].w * slope);
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

    auto compute = [&](int buf) {
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
    };

    // ---- main loop: register sets R0/R1 hold chunks c+1 / c+2 while chunk c is computed from LDS
    const int nchunks = (p.K + BK - 1) / BK;
    float4 ra0[AL], rb0[BL], ra1[AL], rb1[BL];
    if constexpr (!GENERIC) set_tap(0);
    load_a(ra0, 0);
    load_b(rb0, 0);
    load_a(ra1, BK);
    load_b(rb1, BK);
    store_tiles(ra0, rb0, 0);
    __syncthreads();
    // The loop body is branch-free: chunks past K load zeros (and add nothing), so an odd chunk count costs
    // one empty MFMA block instead of a data-dependent exit in the middle of the pipeline.
    for (int c = 0; c < nchunks; c += 2) {
        // even step: chunk c in LDS buf 0, chunk c+1 in R1, R0 free -> prefetch chunk c+2
        load_a(ra0, (c + 2) * BK);
        load_b(rb0, (c + 2) * BK);
        compute(0);
        store_tiles(ra1, rb1, 1);
        __syncthreads();
        // odd step: chunk c+1 in LDS buf 1, chunk c+2 in R0, R1 free -> prefetch chunk c+3
        load_a(ra1, (c + 3) * BK);
        load_b(rb1, (c + 3) * BK);
        compute(1);
        store_tiles(ra0, rb0, 0);
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
    // (all global reads of a block before its first store: one memory round trip per block, see igemm_epilogue.h)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mb = m0 + wm * WTM + i * 32;
        const int b0 = mb / rpb;
        const int nextb = (b0 + 1) * rpb;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WTN + j * 32 + lrow;
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
                if (p.accumulate) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cv[r] = cp[mc[r] * p.ldc + n];
                }
                asm volatile("" : "+v"(bias));      // (loads waited for once, here: see settle() in igemm_epilogue.h)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    asm volatile("" : "+v"(ra[r]));
                    asm volatile("" : "+v"(rs[r]));
                    asm volatile("" : "+v"(cv[r]));
                }
                float outv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * p.alpha + bias;
                    if (p.rowadd) v += ra[r];
                    if (resp) v += rs[r];
                    if (p.act == 1) v = tanhf(v);
                    else if (p.act == 2) v = fmaxf(v, 0.f);
                    else if (p.act == 3) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
                    v *= p.out_scale;
                    if (p.accumulate) v += cv[r];
                    outv[r] = v;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (m < p.M) cp[(long long)m * p.ldc + n] = outv[r];
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(const Ctx& ctx, const IGemm& p, bool generic, int Nb) {
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N * (p.geglu ? 2 : 1) + BN - 1) / BN;
    dim3 grid((unsigned)((long long)mtiles * ntiles), (unsigned)p.Z);
    if (generic && p.b_nk) {
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, true, true>), grid, dim3(NT), 0, ctx.stream, p, ntiles, Nb);
    } else if (generic) {
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, false, true>), grid, dim3(NT), 0, ctx.stream, p, ntiles, Nb);
    } else if (p.b_nk) {
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, true, false>), grid, dim3(NT), 0, ctx.stream, p, ntiles, Nb);
    } else {
        hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WGM, WGN, false, false>), grid, dim3(NT), 0, ctx.stream, p, ntiles, Nb);
    }
}

inline double tile_cost(long long M, long long N, int Z, int BM, int BN, double eff) {
    const long long blocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * Z;
    const long long rounds = (blocks + 255) / 256;
    return (double)rounds * BM * BN / eff;
}

}  // namespace

void launch_igemm(const Ctx& ctx, const IGemm& p_in) {
    IGemm p = p_in;
    p.zeros = ctx.zeros;
    p.no_pair = 0;
    MAA_CHECK(p.zeros != nullptr, "context has no zero page");
    // precision mode of the context: 1 = bf16x3 split, 2 = plain bf16 operands; problems the bf16 engine cannot
    // take (B not k-contiguous, odd channel counts) run on the exact-fp32 kernel below.  (launch_igemm_bf16 also runs
    // in the workspace dry run: its split-K slabs come from the arena.)
    if (ctx.dtype == 1 && launch_igemm_bf16(ctx, p, 3)) return;
    if (ctx.dtype == 2 && launch_igemm_bf16(ctx, p, 1)) return;
    if (ctx.ws.dry) return;
    MAA_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "empty igemm");
    MAA_CHECK(!p.c_split, "split32 output asked of a problem only the fp32 engine can take");
    const int taps = p.KH * p.KW, Ctot = p.C1 + p.C2;
    MAA_CHECK(p.K <= taps * Ctot && p.K > (taps - 1) * Ctot, "igemm K mismatch");
    MAA_CHECK(p.a_act == 0 || p.a_act == 1, "igemm A activation");
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    // fast gather: 16-channel chunks never straddle a tap or a source, float4 loads are aligned
    // (single source, K % 4 != 0: the last float4 over-reads up to 3 floats of the same row, which must exist
    //  and be finite -- e.g. the zeroed padding columns of the attention scores; they meet zero rows of B)
    bool fast = (taps == 1 ? (p.C2 == 0 ? (Ctot % 4 == 0 || p.lda1 >= (Ctot + 3) / 4 * 4)
                                        : (p.C1 % BK == 0 && Ctot % 4 == 0))
                           : (Ctot % BK == 0 && p.C1 % BK == 0)) &&
                p.lda1 % 4 == 0 && al16(p.a1) && p.a_so % 4 == 0 && p.a_si % 4 == 0;
    if (p.C2 > 0) fast = fast && p.lda2 % 4 == 0 && al16(p.a2);
    if (p.Z > 1) MAA_CHECK(p.C2 == 0, "batched igemm takes one A source");
    MAA_CHECK(fast || p.K == taps * Ctot, "padded K needs the aligned gather");
    // columns that may be read from B: packed weights are zero-padded to a multiple of 32
    const int ncols = p.N * (p.geglu ? 2 : 1);
    int Nb = ncols;
    MAA_CHECK(p.ldb % 4 == 0 && al16(p.b) && p.b_so % 4 == 0 && p.b_si % 4 == 0, "B operand alignment");
    if (!p.b_nk) {
        Nb = (ncols + 3) / 4 * 4;
        if (Nb > p.ldb) Nb = p.ldb / 4 * 4;
    } else {
        MAA_CHECK(p.ldb >= (p.K + 3) / 4 * 4, "B [N][K] rows must be padded to a multiple of 4");
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
        cfg = choose_tile(p.M, ncols, p.Z, false);
    }
    static const char* kNames[4] = {"igemm_f32<128x128>", "igemm_f32<128x64>", "igemm_f32<64x64>", "igemm_f32<256x32>"};
    char shape_name[48];
    const char* pname = kNames[cfg];
    if (ctx.prof && ctx.prof->detail) {
        std::snprintf(shape_name, sizeof(shape_name), "ig%d M%d N%d K%d t%d Z%d%s", cfg, p.M, ncols, p.K, taps, p.Z,
                      p.b_nk ? "T" : "");
        pname = shape_name;
    }
    ProfScope prof(ctx, pname, flops, bytes);
    switch (cfg) {
        case 0: launch_cfg<128, 128, 2, 2>(ctx, p, !fast, Nb); break;
        case 1: launch_cfg<128, 64, 2, 2>(ctx, p, !fast, Nb); break;
        case 2: launch_cfg<64, 64, 2, 2>(ctx, p, !fast, Nb); break;
        default: launch_cfg<256, 32, 4, 1>(ctx, p, !fast, Nb); break;
    }
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
