// Box calibration for bench.py's `box.calib`: two fixed, data-independent loops timed on the context's stream
// right before the benchmark's timed region, so that a slow or fast box can be told from a slow or fast build of the engines:
//   kind 0  back-to-back v_mfma_f32_32x32x16_bf16 on eight independent accumulators per wave, 16 waves per CU -> TFLOP/s
//           (the dense bf16 MFMA rate the box sustains at the clock its power limit allows; 2500 at 2.4 GHz)
//   kind 1  float4 copy of 256 MiB -> GB/s read + written (HBM3E: 8 TB/s peak, ~6.3 TB/s for this pattern)
//   kind 2  every workgroup re-reads its own 64 KiB (16 MiB in all: inside the 8 x 4 MiB of L2, past the L1) -> GB/s out of L2
//   kind 3  128 MiB re-read by the whole grid (past L2, inside the 256 MiB Infinity Cache) -> GB/s out of the fabric side
//           (kinds 2 and 3, round 4: the boxes that run the short-K / slab kernels 1.5-2x slower have kinds 0 and 1 normal)
// Not part of the reference's interface; nothing in the product path calls it.
#include "maa_internal.h"

namespace maa {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__global__ __launch_bounds__(256) void calib_mfma_kernel(int iters, float* sink) {
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * (float)((threadIdx.x + i) & 7));
        b[i] = (__bf16)(0.002f * (float)((threadIdx.x + 3 * i) & 7));
    }
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][15];
    if (s == 12345.678f) sink[0] = s;      // (never true: keeps the loop alive)
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

// sum of float4s read from `src`: a workgroup walks its own `per_wg4` float4s `reps` times (kind 2), or the whole grid walks
// n4 float4s grid-stride `reps` times (kind 3: per_wg4 = 0)
__global__ __launch_bounds__(1024) void calib_read_kernel(const float4* __restrict__ src, long long n4, int per_wg4, int reps,
                                                          float* sink) {
    float4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        if (per_wg4) {
            const float4* p = src + (long long)blockIdx.x * per_wg4;
            for (int i = threadIdx.x; i < per_wg4; i += blockDim.x) {
                const float4 v = p[i];
                s.x += v.x;
                s.y += v.y;
                s.z += v.z;
                s.w += v.w;
            }
        } else {
            for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
                const float4 v = src[i];
                s.x += v.x;
                s.y += v.y;
                s.z += v.z;
                s.w += v.w;
            }
        }
        __syncthreads();
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;      // (never true)
}

}  // namespace

double calib_run(const Ctx& ctx, int kind) {
    hipEvent_t e0, e1;
    MAA_HIP(hipEventCreate(&e0));
    MAA_HIP(hipEventCreate(&e1));
    float ms = 0.f;
    double value = 0.0;
    const int cus = device_cu_count(ctx.device);
    if (kind == 0) {
        const int iters = 4096, blocks = 4 * cus, reps = 4;
        hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, ctx.stream, 64, ctx.zeros);      // warm-up
        MAA_HIP(hipEventRecord(e0, ctx.stream));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, ctx.stream, iters, ctx.zeros);
        MAA_HIP(hipEventRecord(e1, ctx.stream));
        MAA_HIP(hipEventSynchronize(e1));
        MAA_HIP(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)reps * blocks * 4.0 * iters * 8.0 * (2.0 * 32 * 32 * 16);
        value = flops / (ms * 1e-3) / 1e12;
    } else if (kind == 2 || kind == 3) {
        const size_t bytes = kind == 2 ? (size_t)cus * (64 << 10) : (size_t)128 << 20;
        const int reps = kind == 2 ? 256 : 16;
        const int per_wg4 = kind == 2 ? (64 << 10) / 16 : 0;
        const int blocks = kind == 2 ? cus : 2 * cus;
        void* src = nullptr;
        MAA_HIP(hipMalloc(&src, bytes));
        MAA_HIP(hipMemsetAsync(src, 0, bytes, ctx.stream));
        const long long n4 = (long long)(bytes / 16);
        hipLaunchKernelGGL(calib_read_kernel, dim3(blocks), dim3(1024), 0, ctx.stream, (const float4*)src, n4, per_wg4, 2, ctx.zeros);
        MAA_HIP(hipEventRecord(e0, ctx.stream));
        hipLaunchKernelGGL(calib_read_kernel, dim3(blocks), dim3(1024), 0, ctx.stream, (const float4*)src, n4, per_wg4, reps, ctx.zeros);
        MAA_HIP(hipEventRecord(e1, ctx.stream));
        MAA_HIP(hipEventSynchronize(e1));
        MAA_HIP(hipEventElapsedTime(&ms, e0, e1));
        MAA_HIP(hipFree(src));
        value = (double)bytes * reps / (ms * 1e-3) / 1e9;
    } else {
        const size_t bytes = (size_t)256 << 20;
        const int reps = 8;
        void *src = nullptr, *dst = nullptr;
        MAA_HIP(hipMalloc(&src, bytes));
        MAA_HIP(hipMalloc(&dst, bytes));
        MAA_HIP(hipMemsetAsync(src, 0x3c, bytes, ctx.stream));
        const long long n4 = (long long)(bytes / 16);
        hipLaunchKernelGGL(calib_copy_kernel, dim3(16 * cus), dim3(256), 0, ctx.stream, (const float4*)src, (float4*)dst, n4);
        MAA_HIP(hipEventRecord(e0, ctx.stream));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(calib_copy_kernel, dim3(16 * cus), dim3(256), 0, ctx.stream, (const float4*)src, (float4*)dst, n4);
        MAA_HIP(hipEventRecord(e1, ctx.stream));
        MAA_HIP(hipEventSynchronize(e1));
        MAA_HIP(hipEventElapsedTime(&ms, e0, e1));
        MAA_HIP(hipFree(src));
        MAA_HIP(hipFree(dst));
        value = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    MAA_HIP(hipGetLastError());
    return value;
}

}  // namespace maa
