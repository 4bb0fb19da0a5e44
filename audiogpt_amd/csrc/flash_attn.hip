// Fused attention for the UNet's multi-head layers: softmax(scale * Q K^T) V without materialising the scores.
//   reference: CrossAttention.forward, ldm/modules/attention.py:170-193 (self: 780 or 195 tokens, d = 40 / 80;
//   cross: 77 or 1 context tokens); QKVAttentionLegacy, openaimodel.py:356-372 (inpaint, 1060 / 265 tokens).
// Used in the bf16 precision modes (operands split into bf16 hi + lo, fp32 accumulation, fp32 softmax); the
// exact-fp32 mode and the single-head VAE blocks (d = 256 / 512) keep the three-launch path in blocks.cpp.
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.  Keys / values
// stream through LDS in tiles of 32 keys shared by the four waves (register prefetch of the next tile, one
// barrier per tile).  Per tile and wave, on the 32x32x16 bf16 MFMA:
//   S^T[key][q]  = K_tile . Q^T        A = K rows (k = head dim, contiguous), B = Q rows held in registers
//   online softmax over keys, per query = per LANE: a lane holds 16 of the 32 keys of its query column, the other
//                  16 sit in lane^32 -> the row max / sum are 15 in-lane ops + one cross-lane exchange
//   O^T[d][q]   += V^T . P^T           A = V^T fragment (LDS holds V transposed), B = the lane's own P registers:
//                  the k-slot -> key map of the MFMA is free as long as A and B agree, so P never moves between
//                  lanes; the running rescale exp(m_old - m_new) is one scalar per lane.
// O^T is transposed through LDS at the end so that the output rows are written as contiguous runs of d.
#include "maa_internal.h"

#include <cstdlib>

namespace maa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FlashArgs {
    const float *q, *k, *v;
    int ldq, ldk, ldv;            // row pitch (floats)
    int hsq, hsk, hsv;            // per-head column offset
    long long q_bs, k_bs, v_bs;   // per-batch offset
    int heads, Nq, Nk;
    float scale;
    float* out;
    int ldo;
    int out_split;                // write split32 lines (the to_out projection reads them with no conversion)
    int causal;                   // query i sees keys 0 .. i only (OpenCLIP's text tower)
    int qtiles, xcd_on;           // 128-query tiles per (sample, head); XCD-contiguous work order (maa_internal.h)
    long long o_bs;
    const float* zeros;
};

namespace {

constexpr int NT = 256;
constexpr int KT = 32;      // keys per tile

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pk_bf16(a, b);
    lo = pk_bf16(a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u));
}
struct Frag {      // 8 bf16 = 4 dwords, bit-castable to the MFMA operand type
    unsigned w[4];
};
__device__ __forceinline__ bf16x8 as_bf16x8(const Frag& f) { return __builtin_bit_cast(bf16x8, f); }

// (capping the d = 40 kernel at 128 VGPRs -- four workgroups per CU -- spills and is 20 % slower: DESIGN.md 3.3)
// (round 6: five-wave workgroups of 160 query rows where that takes fewer waves -- 780 tokens: 5 x 5 waves instead of 7 x 4 with
// three of them empty, each K / V tile staged for five waves -- measured 0.6 % SLOWER end to end, two workgroups per CU instead of
// three: profiles/r6_call12_flash5_ab.txt)
template <int DH, int TERMS>
__global__ __launch_bounds__(NT, 1) void flash_attn_kernel(const FlashArgs a) {
    constexpr int DK = (DH + 15) / 16 * 16;      // head dim padded to the MFMA k-step
    constexpr int DM = (DH + 31) / 32 * 32;      // rows of O^T
    constexpr int KS = DK / 16, MB = DM / 32;
    constexpr int PL = TERMS == 1 ? 1 : 2;
    constexpr int LDKK = DK + 8;                 // bf16 per K-tile row
    constexpr int LDV = KT + 4;                  // bf16 per V^T row: 72 bytes -- the 32 rows a ds_read_b64 group touches start in
                                                 // 32 different even banks (80-byte rows collide two by two)
    constexpr int K_PLANE = KT * LDKK, V_PLANE = DM * LDV;
    constexpr int BUF = PL * (K_PLANE + V_PLANE);            // bf16 elements of one K/V buffer
    constexpr int C4 = DH / 4;                               // float4 per K / V row
    static_assert(DH % 8 == 0, "head dim");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];   // [2][BUF]; reused for the O transpose

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    // work item = ((sample, head), query tile), sample-major; XCD-contiguous: the workgroups of a sample share one XCD, where the
    // q / k / v rows of that sample were written by the projection's tiles and its K / V tiles are fetched into L2 once
    const int w = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, a.xcd_on);
    const int bh = w / a.qtiles, b = bh / a.heads, h = bh - b * a.heads;
    const int q0 = (w - bh * a.qtiles) * 128 + wid * 32;
    const bool wave_live = q0 < a.Nq;          // wave-uniform
    const float* qp = a.q + b * a.q_bs + h * a.hsq;
    const float* kp = a.k + b * a.k_bs + h * a.hsk;
    const float* vp = a.v + b * a.v_bs + h * a.hsv;
    const float4* zero4 = reinterpret_cast<const float4*>(a.zeros);

    // ---- zero both LDS buffers once: the padding (d >= DH rows / columns, keys of a short last tile) must be 0
    for (int i = tid; i < 2 * BUF / 2; i += NT) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- Q^T operand of this wave: lane (q = lq, half lh) holds d = ks*16 + lh*8 .. +7 for every k-step
    Frag qh[KS], ql[KS];
    {
        const int qrow = q0 + lq;
        const bool qok = qrow < a.Nq;
        const float* src = qp + (long long)(qok ? qrow : 0) * a.ldq;
        // the softmax scale and log2 e ride on Q: the scores leave the MFMA ready for v_exp_f32 (2^x)
        const float qs = a.scale * 1.4426950408889634f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 16 + lh * 8;
            float4 x0 = *((qok && d < DH) ? reinterpret_cast<const float4*>(src + d) : zero4);
            float4 x1 = *((qok && d + 4 < DH) ? reinterpret_cast<const float4*>(src + d + 4) : zero4);
            x0.x *= qs;
            x0.y *= qs;
            x0.z *= qs;
            x0.w *= qs;
            x1.x *= qs;
            x1.y *= qs;
            x1.z *= qs;
            x1.w *= qs;
            if constexpr (TERMS == 1) {
                qh[ks].w[0] = pk_bf16(x0.x, x0.y);
                qh[ks].w[1] = pk_bf16(x0.z, x0.w);
                qh[ks].w[2] = pk_bf16(x1.x, x1.y);
                qh[ks].w[3] = pk_bf16(x1.z, x1.w);
            } else {
                split2(x0.x, x0.y, qh[ks].w[0], ql[ks].w[0]);
                split2(x0.z, x0.w, qh[ks].w[1], ql[ks].w[1]);
                split2(x1.x, x1.y, qh[ks].w[2], ql[ks].w[2]);
                split2(x1.z, x1.w, qh[ks].w[3], ql[ks].w[3]);
            }
        }
    }

    // ---- K / V tile staging, spread over all 256 threads.
    // K: element idx = tid + 256 j -> (key, 4 d), one row-contiguous 8-byte store per plane.
    // V: block idx = (255 - tid) + 256 j -> (4 keys kb, 2 d c2): the two d of the four keys are transposed in the registers
    // they were loaded into -- V^T[d][4 keys] is ONE 8-byte store per d and plane (the 2-byte scattered stores this replaces
    // were 5-way bank conflicts: 54 % of the kernel's LDS cycles, profiles/r2_pmc_sq_counters_dma2_first_policy.txt).  The V
    // blocks are dealt from the last thread down and the second pass of K from the first thread up, so no thread gets both
    // extras (d = 40: threads 0-63 two K elements, 96-255 one K element + one V block).
    constexpr int C2 = DH / 2;
    constexpr int NK = KT * C4, NV = (KT / 4) * C2;
    constexpr int NLK = (NK + NT - 1) / NT, NLV = (NV + NT - 1) / NT;
    float4 rk[NLK];
    float2 rv[NLV][4];
    const float2* zero2 = reinterpret_cast<const float2*>(a.zeros);
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NLK; ++j) {
            const int idx = tid + NT * j;
            const int key = idx / C4, c4 = idx - key * C4;
            const int kg = kt * KT + key;
            const bool ok = idx < NK && kg < a.Nk;
            rk[j] = *(ok ? reinterpret_cast<const float4*>(kp + (long long)kg * a.ldk + c4 * 4) : zero4);
        }
#pragma unroll
        for (int j = 0; j < NLV; ++j) {
            const int idx = (NT - 1 - tid) + NT * j;
            const int kb = idx / C2, c2 = idx - kb * C2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kg = kt * KT + kb * 4 + k;
                const bool ok = idx < NV && kg < a.Nk;
                rv[j][k] = *(ok ? reinterpret_cast<const float2*>(vp + (long long)kg * a.ldv + c2 * 2) : zero2);
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned short* base = smem + buf * BUF;
        unsigned short* vt = base + PL * K_PLANE;
#pragma unroll
        for (int j = 0; j < NLK; ++j) {
            const int idx = tid + NT * j;
            if (idx < NK) {
                const int key = idx / C4, c4 = idx - key * C4;
                uint2 hi, lo;
                if constexpr (TERMS == 1) {
                    hi.x = pk_bf16(rk[j].x, rk[j].y);
                    hi.y = pk_bf16(rk[j].z, rk[j].w);
                } else {
                    split2(rk[j].x, rk[j].y, hi.x, lo.x);
                    split2(rk[j].z, rk[j].w, hi.y, lo.y);
                    *reinterpret_cast<uint2*>(base + K_PLANE + key * LDKK + c4 * 4) = lo;
                }
                *reinterpret_cast<uint2*>(base + key * LDKK + c4 * 4) = hi;
            }
        }
#pragma unroll
        for (int j = 0; j < NLV; ++j) {
            const int idx = (NT - 1 - tid) + NT * j;
            if (idx < NV) {
                const int kb = idx / C2, c2 = idx - kb * C2;
                // V^T[d = 2 c2 + i][keys 4 kb .. 4 kb + 3]
                auto put = [&](int d, float v0, float v1, float v2, float v3) {
                    uint2 hi, lo;
                    if constexpr (TERMS == 1) {
                        hi.x = pk_bf16(v0, v1);
                        hi.y = pk_bf16(v2, v3);
                    } else {
                        split2(v0, v1, hi.x, lo.x);
                        split2(v2, v3, hi.y, lo.y);
                        *reinterpret_cast<uint2*>(vt + V_PLANE + d * LDV + kb * 4) = lo;
                    }
                    *reinterpret_cast<uint2*>(vt + d * LDV + kb * 4) = hi;
                };
                put(2 * c2, rv[j][0].x, rv[j][1].x, rv[j][2].x, rv[j][3].x);
                put(2 * c2 + 1, rv[j][0].y, rv[j][1].y, rv[j][2].y, rv[j][3].y);
            }
        }
    };

    f32x16 oacc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[mb][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (a.Nk + KT - 1) / KT;
    load_tile(0);
    __syncthreads();          // zero fill complete
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        load_tile(kt + 1);                         // past the end: every element masked -> zeros
        const unsigned short* base = smem + buf * BUF;
        const unsigned short* vt = base + PL * K_PLANE;

        if (wave_live) {      // a wave whose 32 query rows all lie past Nq (the 780 = 6 x 128 + 12 tail) only helps stage K / V
            // ---- S^T = K_tile . Q^T
            f32x16 sacc;
    #pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    #pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(base + lq * LDKK + ks * 16 + lh * 8);
                if constexpr (TERMS == 3) {
                    const bf16x8 kl = *reinterpret_cast<const bf16x8*>(base + K_PLANE + lq * LDKK + ks * 16 + lh * 8);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, as_bf16x8(qh[ks]), sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, as_bf16x8(ql[ks]), sacc, 0, 0, 0);
                }
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, as_bf16x8(qh[ks]), sacc, 0, 0, 0);
            }

            // ---- online softmax for query lq: this lane's keys are kt*32 + (r&3) + 8*(r>>2) + 4*lh.  The scores are already
            // scaled, in log2 units; only a short last tile or a causal
            // mask needs the per-key test (uniform branch)
            float p[16];
            float mt = -INFINITY;
            if (!a.causal && (kt + 1) * KT <= a.Nk) {
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = sacc[r];
                    mt = fmaxf(mt, p[r]);
                }
            } else {
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KT + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    p[r] = (key < a.Nk && (!a.causal || key <= q0 + lq)) ? sacc[r] : -INFINITY;
                    mt = fmaxf(mt, p[r]);
                }
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            // exp through v_exp_f32 (~1 ulp): the softmax is the VALU-heavy part of this kernel; the running maximum settles
            // after the first tiles, so the rescale of O is skipped for a wave whose lanes all kept their maximum (alpha == 1
            // exactly)
            float alpha, ls = 0.f;
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);                  // first tile: exp2(-inf) = 0
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(p[r] - m_new);                // masked keys: exp2(-inf) = 0
                ls += p[r];
            }
            ls += __shfl_xor(ls, 32, 64);
            l_run = l_run * alpha + ls;
            const bool moved = m_new != m_run;
            m_run = m_new;
            if (__any(moved)) {
    #pragma unroll
                for (int mb = 0; mb < MB; ++mb)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[mb][r] *= alpha;
            }

            // ---- O^T += V^T . P^T ; k-slot (lh*8 + t) of step u  <->  key 16u + 4 lh + (t&3) + 8 (t>>2)  <->  p[8u + t]
    #pragma unroll
            for (int u = 0; u < 2; ++u) {
                Frag ph, pl;
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (TERMS == 1)
                        ph.w[t] = pk_bf16(p[8 * u + 2 * t], p[8 * u + 2 * t + 1]);
                    else
                        split2(p[8 * u + 2 * t], p[8 * u + 2 * t + 1], ph.w[t], pl.w[t]);
                }
    #pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const unsigned short* vrow = vt + (mb * 32 + lq) * LDV + 16 * u + 4 * lh;
                    Frag vh, vl;
                    const uint2 a0 = *reinterpret_cast<const uint2*>(vrow);
                    const uint2 a1 = *reinterpret_cast<const uint2*>(vrow + 8);
                    vh.w[0] = a0.x;
                    vh.w[1] = a0.y;
                    vh.w[2] = a1.x;
                    vh.w[3] = a1.y;
                    if constexpr (TERMS == 3) {
                        const uint2 c0 = *reinterpret_cast<const uint2*>(vrow + V_PLANE);
                        const uint2 c1 = *reinterpret_cast<const uint2*>(vrow + V_PLANE + 8);
                        vl.w[0] = c0.x;
                        vl.w[1] = c0.y;
                        vl.w[2] = c1.x;
                        vl.w[3] = c1.y;
                        oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vl), as_bf16x8(ph), oacc[mb], 0, 0, 0);
                        oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vh), as_bf16x8(pl), oacc[mb], 0, 0, 0);
                    }
                    oacc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vh), as_bf16x8(ph), oacc[mb], 0, 0, 0);
                }
            }

        }
        store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- normalise and transpose O^T through LDS (wave-private 32 x (DM+1) floats), then write rows of d
    float* ot = reinterpret_cast<float*>(smem) + wid * 32 * (DM + 1);
    const float inv = 1.f / l_run;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            ot[lq * (DM + 1) + d] = oacc[mb][r] * inv;
        }
    __syncthreads();
    float* op = a.out + b * a.o_bs + h * DH;
    if (a.out_split) {
        // two neighbouring d per lane: one 4-byte store for the bf16 hi pair and one for the lo pair (h DH and d are even)
        for (int i = lane; i < 16 * DH; i += 64) {
            const int qi = i / (DH / 2), d = (i - qi * (DH / 2)) * 2;
            if (q0 + qi < a.Nq) {
                const float v0 = ot[qi * (DM + 1) + d], v1 = ot[qi * (DM + 1) + d + 1];
                unsigned hi, lo;
                split2(v0, v1, hi, lo);
                const int c = h * DH + d;
                unsigned short* o = reinterpret_cast<unsigned short*>(a.out + b * a.o_bs + (long long)(q0 + qi) * a.ldo) +
                                    (c >> 5) * 64 + (c & 31);
                *reinterpret_cast<unsigned*>(o) = hi;
                *reinterpret_cast<unsigned*>(o + 32) = lo;
            }
        }
    } else {
        for (int i = lane; i < 32 * DH; i += 64) {
            const int qi = i / DH, d = i - qi * DH;
            if (q0 + qi < a.Nq) op[(long long)(q0 + qi) * a.ldo + d] = ot[qi * (DM + 1) + d];
        }
    }
}

template <int DH, int TERMS>
void launch_dh(const Ctx& ctx, const FlashArgs& a, int B) {
    constexpr int DK = (DH + 15) / 16 * 16, DM = (DH + 31) / 32 * 32, PL = TERMS == 1 ? 1 : 2;
    constexpr size_t kv = (size_t)2 * PL * (KT * (DK + 8) + DM * (KT + 4)) * sizeof(unsigned short);
    constexpr size_t tr = (size_t)4 * 32 * (DM + 1) * sizeof(float);
    constexpr size_t lds = kv > tr ? kv : tr;
    auto kern = flash_attn_kernel<DH, TERMS>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), ctx.device, (int)lds);
    dim3 grid((unsigned)(a.qtiles * B * a.heads));
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx.stream, a);
}

}  // namespace

bool flash_attention_covers(const Ctx& ctx, int dh) {
    return ctx.dtype != 0 && (dh == 32 || dh == 40 || dh == 64 || dh == 80);
}

// false: shape not covered (head dim other than 32 / 40 / 64 / 80, unaligned rows) -> caller uses the GEMM path
bool launch_flash_attention(const Ctx& ctx, const float* q, int ldq, int hsq, const float* k, int ldk, int hsk,
                            const float* v, int ldv, int hsv, int B, int heads, int dh, int Nq, int Nk, float alpha,
                            float* out, int ldo, int out_split, int causal) {
    if (!flash_attention_covers(ctx, dh)) return false;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (ldq % 4 || ldk % 4 || ldv % 4 || hsq % 4 || hsk % 4 || hsv % 4 || !al16(q) || !al16(k) || !al16(v)) return false;
    if (ctx.ws.dry) return true;
    FlashArgs a;
    a.q = q;
    a.k = k;
    a.v = v;
    a.ldq = ldq;
    a.ldk = ldk;
    a.ldv = ldv;
    a.hsq = hsq;
    a.hsk = hsk;
    a.hsv = hsv;
    a.q_bs = (long long)Nq * ldq;
    a.k_bs = (long long)Nk * ldk;
    a.v_bs = (long long)Nk * ldv;
    a.heads = heads;
    a.Nq = Nq;
    a.Nk = Nk;
    a.scale = alpha;
    a.out = out;
    a.ldo = ldo;
    a.out_split = out_split;
    a.causal = causal;
    a.o_bs = (long long)Nq * ldo;
    a.zeros = ctx.zeros;
    a.qtiles = (Nq + 127) / 128;
    a.xcd_on = 1;
    const double flops = 4.0 * B * heads * (double)Nq * Nk * dh;
    const double bytes = 4.0 * B * heads * ((double)2 * Nq * dh + 2.0 * Nk * dh);
    ProfScope prof(ctx, "flash_attention", flops, bytes);
    const int terms = ctx.dtype == 1 ? 3 : 1;
#define MAA_FLASH(DHV)                                 \
    if (terms == 3)                                    \
        launch_dh<DHV, 3>(ctx, a, B);                  \
    else                                               \
        launch_dh<DHV, 1>(ctx, a, B);
    switch (dh) {
        case 32: MAA_FLASH(32) break;
        case 40: MAA_FLASH(40) break;
        case 64: MAA_FLASH(64) break;
        default: MAA_FLASH(80) break;
    }
#undef MAA_FLASH
    MAA_HIP(hipGetLastError());
    return true;
}

}  // namespace maa
