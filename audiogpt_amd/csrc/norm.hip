// GroupNorm(32) [+SiLU], LayerNorm and row softmax for channels-last fp32 activations (gfx950).
//   GroupNorm32 / Normalize: ldm/modules/diffusionmodules/util.py:199-216 (eps 1e-5),
//       ldm/modules/attention.py:76-77 and diffusionmodules/model.py:38-39 (eps 1e-6), swish model.py:33-35
//   LayerNorm: ldm/modules/attention.py:203-205 (eps 1e-5)
//   softmax:   ldm/modules/attention.py:188, openaimodel.py:370, model.py:191
// All three are HBM/L2-bound; reductions are wave64 shuffles plus one LDS hop across the 4 waves.
#include "maa_internal.h"

namespace maa {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 f = {a, b};
    bf16x2 h = __builtin_convertvector(f, bf16x2);
    return __builtin_bit_cast(unsigned, h);
}
// write 4 consecutive channels (c % 4 == 0) of row `row` either as fp32 or in the "split32" form read by the bf16
// GEMM engine: every 32 channels of a row take one 128-byte line [32 bf16 hi | 32 bf16 lo], hi = bf16(v),
// lo = bf16(v - hi); the row pitch is the same C*4 bytes as fp32
__device__ __forceinline__ void store4(float* out, long long row, int c, int C, int split, float4 v) {
    if (!split) {
        *reinterpret_cast<float4*>(out + row * C + c) = v;
        return;
    }
    unsigned short* o = reinterpret_cast<unsigned short*>(out) + row * C * 2 + (c >> 5) * 64 + (c & 31);
    uint2 hi, lo;
    hi.x = pk_bf16(v.x, v.y);
    hi.y = pk_bf16(v.z, v.w);
    lo.x = pk_bf16(v.x - __builtin_bit_cast(float, hi.x << 16), v.y - __builtin_bit_cast(float, hi.x & 0xffff0000u));
    lo.y = pk_bf16(v.z - __builtin_bit_cast(float, hi.y << 16), v.w - __builtin_bit_cast(float, hi.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(o) = hi;
    *reinterpret_cast<uint2*>(o + 32) = lo;
}

// GroupNorm pass 1: one block per (sample, group) -> tab[(b*C + c)*2] = {scale, shift} for the group's channels
// (scale = gamma*rstd, shift = beta - mean*scale, as ATen forms them); threads = (position
// lane, channel-in-group).  One pass over the data: sums of d = x - pivot and d*d, with the group's first element as
// pivot (so the subtraction var = E[d^2] - E[d]^2 cancels at most a couple of bits, like the two-pass form), eight
// independent loads in flight per thread.  Fixed reduction order: results are bit-reproducible.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x1, int ld1, int C1,
                                                       const float* __restrict__ x2, int ld2, int C2, int HW,
                                                       int groups, float eps, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ tab) {
    __shared__ float red[4];
    __shared__ float mr[2];
    const int C = C1 + C2, cpg = C / groups;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int tp_n = 256 / cpg;
    const int tc = threadIdx.x % cpg, tp = threadIdx.x / cpg;
    const bool active = tp < tp_n;
    const int c = g * cpg + tc;
    const float* src;
    long long ld;
    if (c < C1) {
        src = x1 + (long long)b * HW * ld1 + c;
        ld = ld1;
    } else {
        src = x2 + (long long)b * HW * ld2 + (c - C1);
        ld = ld2;
    }
    const int c0 = g * cpg;
    const float pivot = c0 < C1 ? x1[(long long)b * HW * ld1 + c0] : x2[(long long)b * HW * ld2 + (c0 - C1)];
    const float n = (float)HW * (float)cpg;
    // eight independent loads in flight per thread: the pass is latency-bound (a 10 x 78 level is 31 positions per thread)
    constexpr int U = 8;
    float sa[U], qa[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sa[u] = qa[u] = 0.f;
    if (active) {
        for (int pos = tp; pos < HW; pos += U * tp_n) {          // a position past the end reads as the pivot: d = 0
            float d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pu = pos + u * tp_n;
                d[u] = pu < HW ? src[(long long)pu * ld] : pivot;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] -= pivot;
                sa[u] += d[u];
                qa[u] += d[u] * d[u];
            }
        }
    }
    const float s0 = sa[0] + sa[4], s1 = sa[1] + sa[5], s2 = sa[2] + sa[6], s3 = sa[3] + sa[7];
    const float q0 = qa[0] + qa[4], q1 = qa[1] + qa[5], q2 = qa[2] + qa[6], q3 = qa[3] + qa[7];
    const float sm = block_sum((s0 + s1) + (s2 + s3), red) / n;
    const float qm = block_sum((q0 + q1) + (q2 + q3), red) / n;
    if (threadIdx.x == 0) {
        const float var = fmaxf(qm - sm * sm, 0.f);
        mr[0] = pivot + sm;
        mr[1] = 1.f / sqrtf(var + eps);
    }
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
        const int cc = c0 + threadIdx.x;
        const float sc = gamma[cc] * mr[1];
        tab[2 * ((long long)b * C + cc)] = sc;
        tab[2 * ((long long)b * C + cc) + 1] = beta[cc] - mr[0] * sc;
    }
}

// GroupNorm pass 2: y = x*scale + shift [+ SiLU].  Block = 16 rows of one sample x all channels, thread = (row lane,
// channel-quad lane): the per-(sample, channel) scale/shift pair is fetched once per quad and reused down the rows;
// 16 quad lanes = 256 contiguous bytes per row, fp32 or split32 out.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x1, int ld1, int C1,
                                                       const float* __restrict__ x2, int ld2, int C2, int HW,
                                                       const float* __restrict__ tab, int silu,
                                                       float* __restrict__ out, int split, float* __restrict__ raw_split) {
    const int C = C1 + C2, c4n = C >> 2;
    const int b = blockIdx.y, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r0 = blockIdx.x * 16 + ty;
    if (r0 >= HW) return;
    const long long row = (long long)b * HW + r0;
    const float4* t4 = reinterpret_cast<const float4*>(tab + 2 * (long long)b * C);
    for (int q = tx + 16 * blockIdx.z; q < c4n; q += 16 * gridDim.z) {      // grid.z splits the quads of small tensors
        const int c = q * 4;
        const float4 ta = t4[2 * q], tb = t4[2 * q + 1];      // {sc0, sh0, sc1, sh1}, {sc2, sh2, sc3, sh3}
        const float4 v = c < C1 ? *reinterpret_cast<const float4*>(x1 + row * ld1 + c)
                                : *reinterpret_cast<const float4*>(x2 + row * ld2 + (c - C1));
        float4 y;
        y.x = v.x * ta.x + ta.y;
        y.y = v.y * ta.z + ta.w;
        y.z = v.z * tb.x + tb.y;
        y.w = v.w * tb.z + tb.w;
        if (silu) {
            y.x = __fdividef(y.x, 1.f + __expf(-y.x));
            y.y = __fdividef(y.y, 1.f + __expf(-y.y));
            y.z = __fdividef(y.z, 1.f + __expf(-y.z));
            y.w = __fdividef(y.w, 1.f + __expf(-y.w));
        }
        store4(out, row, c, C, split, y);
        // second output: the UN-normalised row as split32 (the ResBlock's 1x1 skip convolution reads the same (h | skip) rows:
        // with this copy both of its operands reach LDS by DMA instead of through the register-staged engine)
        if (raw_split) store4(raw_split, row, c, C, 1, v);
    }
}


// GroupNorm in ONE pass over the data (round 5): a workgroup owns GPB consecutive groups of one sample -- CB = GPB cpg channels,
// a run of CB * 4 contiguous bytes per position -- and keeps all of that sample's rows of them in registers: thread (row lane ty,
// float4 column tx) loads rows ty, ty + RL, ... (NR float4s, all in flight together), the statistics are reduced through LDS in a
// fixed order, and the normalised (+ SiLU) rows are written from the registers -- fp32 or split32, plus the optional raw split32
// copy -- so the tensor is read once and the statistics launch, its table and the dependent second launch are gone (two launches
// of 10 + 8 us per GroupNorm at the UNet's sizes, both latency-bound: DESIGN.md 3.4).  Same arithmetic as the two-pass kernels:
// sums of d = x - pivot and d^2 with the group's first element as pivot, scale = gamma rstd, shift = beta - mean scale,
// y = x scale + shift.  The layout (GPB, RL, NR) is a function of (C, HW) only and a workgroup sees one sample: a sample's result
// does not depend on its batch.
template <int NR>
__global__ __launch_bounds__(1024) void gn_fused_kernel(const float* __restrict__ x1, int ld1, int C1,
                                                        const float* __restrict__ x2, int ld2, int C2, int HW, int groups,
                                                        int GPB, float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu, float* __restrict__ out,
                                                        int split, float* __restrict__ raw_split, int nblk, int xcd_on) {
    extern __shared__ float gsm[];      // red[RL Q 8] | colsum[2 CB] | gstat[2 GPB]
    const int C = C1 + C2, cpg = C / groups, CB = GPB * cpg, Q = CB >> 2;
    const int RL = (int)blockDim.x / Q;                      // (blockDim.x == RL Q)
    // work item = (sample, group block), sample-major; XCD-contiguous: a sample's rows are read and written on the XCD whose
    // igemm tiles produce / consume them
    const int w = xcd_contiguous((int)blockIdx.x, (int)gridDim.x, xcd_on);
    const int b = w / nblk, c_lo = (w - b * nblk) * CB;
    const int t = threadIdx.x, ty = t / Q, tx = t - ty * Q;
    const int c = c_lo + 4 * tx;
    float* const red = gsm;
    float* const colsum = gsm + (size_t)RL * Q * 8;
    float* const gstat = colsum + 2 * CB;
    const float* src;
    long long ld;
    if (c < C1) {
        src = x1 + (long long)b * HW * ld1 + c;
        ld = ld1;
    } else {
        src = x2 + (long long)b * HW * ld2 + (c - C1);
        ld = ld2;
    }
    // pivot of each of the float4's four channels = first element of that channel's group
    int gl[4];
    float piv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gl[j] = (4 * tx + j) / cpg;
        const int c0 = c_lo + gl[j] * cpg;
        piv[j] = c0 < C1 ? x1[(long long)b * HW * ld1 + c0] : x2[(long long)b * HW * ld2 + (c0 - C1)];
    }
    float4 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = ty + RL * k;
        v[k] = r < HW ? *reinterpret_cast<const float4*>(src + (long long)r * ld) : make_float4(piv[0], piv[1], piv[2], piv[3]);
    }
    float sd[4] = {0.f, 0.f, 0.f, 0.f}, qd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const float d0 = v[k].x - piv[0], d1 = v[k].y - piv[1], d2 = v[k].z - piv[2], d3 = v[k].w - piv[3];
        sd[0] += d0;
        sd[1] += d1;
        sd[2] += d2;
        sd[3] += d3;
        qd[0] += d0 * d0;
        qd[1] += d1 * d1;
        qd[2] += d2 * d2;
        qd[3] += d3 * d3;
    }
    {
        float4* r4 = reinterpret_cast<float4*>(red + (size_t)t * 8);
        r4[0] = make_float4(sd[0], sd[1], sd[2], sd[3]);
        r4[1] = make_float4(qd[0], qd[1], qd[2], qd[3]);
    }
    __syncthreads();
    // per channel and statistic: the RL row lanes' partials, four interleaved chains in a fixed order
    if (t < 2 * CB) {
        const int stat = t / CB, cc = t - stat * CB;
        const float* p0 = red + (size_t)(cc >> 2) * 8 + stat * 4 + (cc & 3);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int y = 0;
        for (; y + 3 < RL; y += 4) {
            a0 += p0[(size_t)(y + 0) * Q * 8];
            a1 += p0[(size_t)(y + 1) * Q * 8];
            a2 += p0[(size_t)(y + 2) * Q * 8];
            a3 += p0[(size_t)(y + 3) * Q * 8];
        }
        for (; y < RL; ++y) a0 += p0[(size_t)y * Q * 8];
        colsum[t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (t < GPB) {
        float s = 0.f, q = 0.f;
        for (int i = 0; i < cpg; ++i) {
            s += colsum[t * cpg + i];
            q += colsum[CB + t * cpg + i];
        }
        const int c0 = c_lo + t * cpg;
        const float pv = c0 < C1 ? x1[(long long)b * HW * ld1 + c0] : x2[(long long)b * HW * ld2 + (c0 - C1)];
        const float n = (float)HW * (float)cpg;
        const float sm = s / n, qm = q / n;
        const float var = fmaxf(qm - sm * sm, 0.f);
        gstat[2 * t] = pv + sm;
        gstat[2 * t + 1] = 1.f / sqrtf(var + eps);
    }
    __syncthreads();
    float sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = gamma[c + j] * gstat[2 * gl[j] + 1];
        sh[j] = beta[c + j] - gstat[2 * gl[j]] * sc[j];
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = ty + RL * k;
        if (r < HW) {
            const long long row = (long long)b * HW + r;
            float4 y;
            y.x = v[k].x * sc[0] + sh[0];
            y.y = v[k].y * sc[1] + sh[1];
            y.z = v[k].z * sc[2] + sh[2];
            y.w = v[k].w * sc[3] + sh[3];
            if (silu) {
                y.x = __fdividef(y.x, 1.f + __expf(-y.x));
                y.y = __fdividef(y.y, 1.f + __expf(-y.y));
                y.z = __fdividef(y.z, 1.f + __expf(-y.z));
                y.w = __fdividef(y.w, 1.f + __expf(-y.w));
            }
            store4(out, row, c, C, split, y);
            if (raw_split) store4(raw_split, row, c, C, 1, v[k]);
        }
    }
}

// Layout of the one-pass GroupNorm for a tensor of C channels x HW positions per sample, or gpb = 0 when a sample's rows of
// even one group block do not fit the registers of a workgroup (the VAE's large images: the two-pass kernels stay).  A function
// of (C, HW, groups) only.
struct GnFusedPlan {
    int gpb = 0, nr = 0, threads = 0;
    size_t lds = 0;
};
GnFusedPlan gn_fused_plan(int C, int C1, int HW, int groups) {
    GnFusedPlan pl;
    const int cpg = C / groups;
    // at most eight float4 rows per thread where some block width allows it (no register pressure at 1024 threads), else
    // twelve, else sixteen (which spills a little); wider blocks (longer contiguous runs per position) first
    for (int cap : {8, 12, 16})
        for (int gpb : {4, 2, 8, 1, 16}) {
            if (groups % gpb) continue;
            const int CB = gpb * cpg;
            if (CB % 4 || CB > 512) continue;
            const int Q = CB / 4, RL = 1024 / Q;
            if (RL < 1) continue;
            const int need = (HW + RL - 1) / RL;
            if (need > cap) continue;
            pl.gpb = gpb;
            pl.nr = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : need <= 12 ? 12 : 16;
            pl.threads = RL * Q;
            pl.lds = ((size_t)RL * Q * 8 + 2 * CB + 2 * gpb) * sizeof(float);
            return pl;
        }
    return pl;
}

// LayerNorm: one wave per row, four channels per lane per step, row cached in registers (C <= 256*NR)
template <int NR>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long long rows, int C,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ out, int split, int xcd_on) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)xcd_contiguous((int)blockIdx.x, (int)gridDim.x, xcd_on) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = x + row * C;
    float4 v[NR];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = (lane + 64 * i) * 4;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) t = *reinterpret_cast<const float4*>(src + c);
        v[i].x = t.x;
        v[i].y = t.y;
        v[i].z = t.z;
        v[i].w = t.w;
        s += (t.x + t.y) + (t.z + t.w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < C) {
            const float a = v[i].x - mean, b = v[i].y - mean, d = v[i].z - mean, e = v[i].w - mean;
            q += (a * a + b * b) + (d * d + e * e);
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < C) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 bt = *reinterpret_cast<const float4*>(beta + c);
            float4 y;
            y.x = (v[i].x - mean) * rstd * g.x + bt.x;
            y.y = (v[i].y - mean) * rstd * g.y + bt.y;
            y.z = (v[i].z - mean) * rstd * g.z + bt.z;
            y.w = (v[i].w - mean) * rstd * g.w + bt.w;
            store4(out, row, c, C, split, y);
        }
    }
}

// one wave per row; NR > 0: row cached in registers, NR == 0: three passes over memory
// causal_nq > 0: row r belongs to query r % causal_nq, which sees keys 0 .. that index only (the rest are written as 0)
template <int NR>
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ s, long long rows, int cols, int ld, int causal_nq) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = s + row * ld;
    if (causal_nq > 0) {
        const int q = (int)(row % causal_nq);
        cols = cols < q + 1 ? cols : q + 1;
    }
    if constexpr (NR > 0) {
        float v[NR];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cols ? p[c] : -INFINITY;
            m = fmaxf(m, v[i]);
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cols ? expf(v[i] - m) : 0.f;
            sum += v[i];
        }
        const float inv = 1.f / wave_sum(sum);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = lane + 64 * i;
            if (c < ld) p[c] = c < cols ? v[i] * inv : 0.f;
        }
    } else {
        float m = -INFINITY;
        for (int c = lane; c < cols; c += 64) m = fmaxf(m, p[c]);
        m = wave_max(m);
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float e = expf(p[c] - m);
            p[c] = e;
            sum += e;
        }
        const float inv = 1.f / wave_sum(sum);
        for (int c = lane; c < ld; c += 64) p[c] = c < cols ? p[c] * inv : 0.f;
    }
}

// fp32 rows -> split32 rows (same pitch): what a normalisation with out_split = 1 does to its result, on its own
// (slope != 1: leaky-relu first -- the pre-activated input of the vocoders' first MRF convolution)
__global__ __launch_bounds__(256) void split32_pack_kernel(const float* __restrict__ x, long long rows, int C, float slope,
                                                           float* __restrict__ out) {
    const long long n4 = rows * (C / 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long row = i / (C / 4);
        const int c = (int)(i - row * (C / 4)) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + row * C + c);
        v.x = v.x > 0.f ? v.x : v.x * slope;
        v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope;
        v.w = v.w > 0.f ? v.w : v.w * slope;
        store4(out, row, c, C, 1, v);
    }
}

// split32 rows -> fp32 rows (hi + lo): tests of the kernels that only emit the split form
__global__ __launch_bounds__(256) void split32_unpack_kernel(const float* __restrict__ x, long long rows, int C, float* __restrict__ out) {
    const long long n = rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long row = i / C;
        const int c = (int)(i - row * C);
        const unsigned short* line = reinterpret_cast<const unsigned short*>(x) + row * C * 2 + (c >> 5) * 64 + (c & 31);
        out[i] = __builtin_bit_cast(float, (unsigned)line[0] << 16) + __builtin_bit_cast(float, (unsigned)line[32] << 16);
    }
}

}  // namespace

void launch_split32_unpack(const Ctx& ctx, const float* x, long long rows, int C, float* out) {
    if (ctx.ws.dry) return;
    MAA_CHECK(C % 32 == 0, "split32 rows are whole 32-channel lines");
    const long long n = rows * C;
    const unsigned grid = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(split32_unpack_kernel, dim3(grid), dim3(256), 0, ctx.stream, x, rows, C, out);
    MAA_HIP(hipGetLastError());
}

void launch_split32_pack(const Ctx& ctx, const float* x, long long rows, int C, float* out, float slope) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "split32_pack_kernel", 0.0, 8.0 * rows * (double)C);
    MAA_CHECK(C % 32 == 0, "split32 rows are whole 32-channel lines");
    const long long n4 = rows * (C / 4);
    const unsigned grid = (unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipLaunchKernelGGL(split32_pack_kernel, dim3(grid), dim3(256), 0, ctx.stream, x, rows, C, slope, out);
    MAA_HIP(hipGetLastError());
}

void launch_groupnorm(Ctx& ctx, const float* x1, int ld1, int C1, const float* x2, int ld2, int C2, int B, int HW,
                      int groups, const float* gamma, const float* beta, float eps, int silu, float* out,
                      int out_split, float* raw_split) {
    const int C = C1 + C2;
    float* tab = ctx.ws.alloc_f((size_t)2 * B * C);    // per-(sample, channel) {scale, shift}; released with the caller's arena mark
    if (ctx.ws.dry) return;
    MAA_CHECK(C % groups == 0 && C / groups <= 256 && C % 4 == 0 && C1 % 4 == 0 && ld1 % 4 == 0 && (C2 == 0 || ld2 % 4 == 0),
              "groupnorm channels");
    ProfScope prof(ctx, "groupnorm", 0.0, 8.0 * B * (double)HW * C);
    const GnFusedPlan fp = ctx.tune.gn_two_pass ? GnFusedPlan() : gn_fused_plan(C, C1, HW, groups);
    if (fp.gpb) {
        const int nblk = groups / fp.gpb;
        dim3 grid((unsigned)(nblk * B));
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, dim3((unsigned)fp.threads), fp.lds, ctx.stream, x1, ld1, C1, x2, ld2, C2, HW, groups, fp.gpb, eps,
                               gamma, beta, silu, out, out_split, raw_split, nblk, 1);
        };
        switch (fp.nr) {
            case 1: go(gn_fused_kernel<1>); break;
            case 2: go(gn_fused_kernel<2>); break;
            case 4: go(gn_fused_kernel<4>); break;
            case 8: go(gn_fused_kernel<8>); break;
            case 12: go(gn_fused_kernel<12>); break;
            default: go(gn_fused_kernel<16>); break;
        }
        MAA_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * groups), dim3(256), 0, ctx.stream, x1, ld1, C1, x2, ld2, C2, HW, groups,
                       eps, gamma, beta, tab);
    const int rb = (HW + 15) / 16, passes = (C / 4 + 15) / 16;
    int nz = (768 + rb * B - 1) / (rb * B);         // aim at >= 3 workgroups per CU
    nz = nz < 1 ? 1 : nz > passes ? passes : nz;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)rb, (unsigned)B, (unsigned)nz), dim3(256), 0, ctx.stream, x1, ld1,
                       C1, x2, ld2, C2, HW, tab, silu, out, out_split, raw_split);
    MAA_HIP(hipGetLastError());
}

void launch_layernorm(const Ctx& ctx, const float* x, long long rows, int C, const float* gamma, const float* beta,
                      float eps, float* out, int out_split) {
    if (ctx.ws.dry) return;
    MAA_CHECK(C <= 2048 && C % 4 == 0, "layernorm width");
    ProfScope prof(ctx, "layernorm", 0.0, 8.0 * rows * (double)C);
    dim3 grid((unsigned)((rows + 3) / 4));
    const int xa = 1;
    if (C <= 256)
        hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, ctx.stream, x, rows, C, gamma, beta, eps, out, out_split, xa);
    else if (C <= 512)
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, ctx.stream, x, rows, C, gamma, beta, eps, out, out_split, xa);
    else if (C <= 1024)
        hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, ctx.stream, x, rows, C, gamma, beta, eps, out, out_split, xa);
    else      // the ViT-H image tower's 1280-wide rows
        hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, ctx.stream, x, rows, C, gamma, beta, eps, out, out_split, xa);
    MAA_HIP(hipGetLastError());
}

void launch_softmax(const Ctx& ctx, float* s, long long rows, int cols, int ld, int causal_nq) {
    if (ctx.ws.dry) return;
    MAA_CHECK(ld >= cols, "softmax ld");
    ProfScope prof(ctx, "softmax", 0.0, 8.0 * rows * (double)ld);
    dim3 grid((unsigned)((rows + 3) / 4));
    if (ld <= 128)
        hipLaunchKernelGGL(softmax_kernel<2>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld, causal_nq);
    else if (ld <= 256)
        hipLaunchKernelGGL(softmax_kernel<4>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld, causal_nq);
    else if (ld <= 1088)
        hipLaunchKernelGGL(softmax_kernel<17>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld, causal_nq);
    else
        hipLaunchKernelGGL(softmax_kernel<0>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld, causal_nq);
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
