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

// one block per (sample, group); threads = (position lane, channel-in-group)
__global__ __launch_bounds__(256) void groupnorm_kernel(const float* __restrict__ x1, int ld1, int C1,
                                                        const float* __restrict__ x2, int ld2, int C2, int HW,
                                                        int groups, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int silu,
                                                        float* __restrict__ out) {
    __shared__ float red[4];
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
    const float n = (float)HW * (float)cpg;
    float s = 0.f;
    if (active)
        for (int pos = tp; pos < HW; pos += tp_n) s += src[pos * ld];
    const float mean = block_sum(s, red) / n;
    float q = 0.f;
    if (active)
        for (int pos = tp; pos < HW; pos += tp_n) {
            const float d = src[pos * ld] - mean;
            q += d * d;
        }
    const float var = block_sum(q, red) / n;
    const float rstd = 1.f / sqrtf(var + eps);
    if (active) {
        const float ga = gamma[c] * rstd, be = beta[c] - mean * rstd * gamma[c];
        float* dst = out + (long long)b * HW * C + c;
        for (int pos = tp; pos < HW; pos += tp_n) {
            float v = src[pos * ld] * ga + be;
            if (silu) v = v / (1.f + expf(-v));
            dst[(long long)pos * C] = v;
        }
    }
}

// one wave per row, row cached in registers (C <= 64*NR)
template <int NR>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int rows, int C,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = x + (long long)row * C;
    float v[NR];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < C ? src[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        const float d = c < C ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
    float* dst = out + (long long)row * C;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + 64 * i;
        if (c < C) dst[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

// one wave per row; NR > 0: row cached in registers, NR == 0: three passes over memory
template <int NR>
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ s, long long rows, int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = s + row * ld;
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

}  // namespace

void launch_groupnorm(const Ctx& ctx, const float* x1, int ld1, int C1, const float* x2, int ld2, int C2, int B,
                      int HW, int groups, const float* gamma, const float* beta, float eps, int silu, float* out) {
    if (ctx.ws.dry) return;
    const int C = C1 + C2;
    MAA_CHECK(C % groups == 0 && C / groups <= 256, "groupnorm channels");
    ProfScope prof(ctx, "groupnorm", 0.0, 8.0 * B * (double)HW * C);
    hipLaunchKernelGGL(groupnorm_kernel, dim3(B * groups), dim3(256), 0, ctx.stream, x1, ld1, C1, x2, ld2, C2, HW,
                       groups, gamma, beta, eps, silu, out);
    MAA_HIP(hipGetLastError());
}

void launch_layernorm(const Ctx& ctx, const float* x, int rows, int C, const float* gamma, const float* beta,
                      float eps, float* out) {
    if (ctx.ws.dry) return;
    MAA_CHECK(C <= 1024, "layernorm width");
    ProfScope prof(ctx, "layernorm", 0.0, 8.0 * rows * (double)C);
    dim3 grid((rows + 3) / 4);
    if (C <= 320)
        hipLaunchKernelGGL(layernorm_kernel<5>, grid, dim3(256), 0, ctx.stream, x, rows, C, gamma, beta, eps, out);
    else if (C <= 640)
        hipLaunchKernelGGL(layernorm_kernel<10>, grid, dim3(256), 0, ctx.stream, x, rows, C, gamma, beta, eps, out);
    else
        hipLaunchKernelGGL(layernorm_kernel<16>, grid, dim3(256), 0, ctx.stream, x, rows, C, gamma, beta, eps, out);
    MAA_HIP(hipGetLastError());
}

void launch_softmax(const Ctx& ctx, float* s, long long rows, int cols, int ld) {
    if (ctx.ws.dry) return;
    MAA_CHECK(ld >= cols, "softmax ld");
    ProfScope prof(ctx, "softmax", 0.0, 8.0 * rows * (double)ld);
    dim3 grid((unsigned)((rows + 3) / 4));
    if (ld <= 128)
        hipLaunchKernelGGL(softmax_kernel<2>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld);
    else if (ld <= 256)
        hipLaunchKernelGGL(softmax_kernel<4>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld);
    else if (ld <= 1088)
        hipLaunchKernelGGL(softmax_kernel<17>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld);
    else
        hipLaunchKernelGGL(softmax_kernel<0>, grid, dim3(256), 0, ctx.stream, s, rows, cols, ld);
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
