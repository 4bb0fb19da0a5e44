// Small kernels of the conditioning encoders (encoders.cpp): everything that is not a GEMM, a LayerNorm or attention.
#include "maa_internal.h"

namespace maa {

namespace {

constexpr int NT = 256;

// BertEmbeddings before its LayerNorm (transformers modeling_bert.py, BertEmbeddings.forward): word_embeddings[id] +
// token_type_embeddings[0] + position_embeddings[position]; ids out of range are clamped (the reference would raise)
__global__ __launch_bounds__(NT) void bert_embed_kernel(const int* __restrict__ ids, int L, int C, int vocab,
                                                         const float* __restrict__ word, const float* __restrict__ pos,
                                                         const float* __restrict__ type0, float* __restrict__ out) {
    const long long row = blockIdx.x;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int l = (int)(row % L);
    const float4* w = reinterpret_cast<const float4*>(word + (long long)id * C);
    const float4* p = reinterpret_cast<const float4*>(pos + (long long)l * C);
    const float4* t = reinterpret_cast<const float4*>(type0);      // null: no token-type term (OpenCLIP's text tower)
    float4* o = reinterpret_cast<float4*>(out + row * C);
    for (int c = threadIdx.x; c < C / 4; c += NT) {
        const float4 a = w[c], b = t ? t[c] : make_float4(0.f, 0.f, 0.f, 0.f), d = p[c];
        o[c] = make_float4((a.x + b.x) + d.x, (a.y + b.y) + d.y, (a.z + b.z) + d.z, (a.w + b.w) + d.w);
    }
}

// open_clip VisionTransformer.forward: x = cat([class_embedding, patches], dim=1) + positional_embedding
__global__ __launch_bounds__(NT) void vit_tokens_kernel(const float* __restrict__ patches, int P, int C,
                                                         const float* __restrict__ cls, const float* __restrict__ pos,
                                                         float* __restrict__ out) {
    const long long row = blockIdx.x;               // over B * (P + 1)
    const int tok = (int)(row % (P + 1));
    const long long b = row / (P + 1);
    const float4* src = reinterpret_cast<const float4*>(tok == 0 ? cls : patches + (b * P + tok - 1) * C);
    const float4* p = reinterpret_cast<const float4*>(pos + (long long)tok * C);
    float4* o = reinterpret_cast<float4*>(out + row * C);
    for (int c = threadIdx.x; c < C / 4; c += NT) {
        const float4 a = src[c], d = p[c];
        o[c] = make_float4(a.x + d.x, a.y + d.y, a.z + d.z, a.w + d.w);
    }
}

// out[b, :] = x[b * stride : b * stride + C]   (the CLS rows of a [B, L, C] sequence)
__global__ __launch_bounds__(NT) void gather_rows_kernel(const float* __restrict__ x, long long stride, int C,
                                                          float* __restrict__ out) {
    const long long b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += NT) out[b * C + c] = x[b * stride + c];
}

// out[b, :] = x[b, argmax_l ids[b, l], :]   (open_clip encode_text: the features at the end-of-text token, which has the
// highest id; first maximum on ties, like torch.argmax)
__global__ __launch_bounds__(NT) void gather_argmax_rows_kernel(const int* __restrict__ ids, int L, const float* __restrict__ x,
                                                                 int C, float* __restrict__ out) {
    __shared__ int best;
    const long long b = blockIdx.x;
    if (threadIdx.x == 0) {
        int bi = 0, bv = ids[b * L];
        for (int l = 1; l < L; ++l) {
            const int v = ids[b * L + l];
            if (v > bv) {
                bv = v;
                bi = l;
            }
        }
        best = bi;
    }
    __syncthreads();
    const float* src = x + (b * L + best) * C;
    for (int c = threadIdx.x; c < C; c += NT) out[b * C + c] = src[c];
}

// z /= z.norm(dim=-1, keepdim=True): one workgroup per row
__global__ __launch_bounds__(NT) void l2norm_rows_kernel(const float* __restrict__ x, int C, float* __restrict__ out) {
    __shared__ float part[NT / 64];
    const long long b = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += NT) {
        const float v = x[b * C + c];
        s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < NT / 64; ++i) tot += part[i];
    const float n = sqrtf(tot);
    for (int c = threadIdx.x; c < C; c += NT) out[b * C + c] = x[b * C + c] / n;
}

// y = gelu(x) (erf form: torch.nn.functional.gelu default)
__global__ __launch_bounds__(NT) void gelu_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float v = x[i];
        out[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    }
}

}  // namespace

void launch_bert_embed(const Ctx& ctx, const int* ids, long long rows, int L, int C, int vocab, const float* word,
                       const float* pos, const float* type0, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "bert_embed_kernel", 0.0, 8.0 * rows * (double)C);
    hipLaunchKernelGGL(bert_embed_kernel, dim3((unsigned)rows), dim3(NT), 0, ctx.stream, ids, L, C, vocab, word, pos, type0, out);
    MAA_HIP(hipGetLastError());
}

void launch_vit_tokens(const Ctx& ctx, const float* patches, int B, int P, int C, const float* cls, const float* pos,
                       float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "vit_tokens_kernel", 0.0, 8.0 * B * (P + 1.0) * C);
    hipLaunchKernelGGL(vit_tokens_kernel, dim3((unsigned)(B * (P + 1))), dim3(NT), 0, ctx.stream, patches, P, C, cls, pos, out);
    MAA_HIP(hipGetLastError());
}

void launch_gather_rows(const Ctx& ctx, const float* x, long long stride, int B, int C, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "gather_rows_kernel", 0.0, 8.0 * B * (double)C);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)B), dim3(NT), 0, ctx.stream, x, stride, C, out);
    MAA_HIP(hipGetLastError());
}

void launch_gather_argmax_rows(const Ctx& ctx, const int* ids, int B, int L, const float* x, int C, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "gather_argmax_rows_kernel", 0.0, 8.0 * B * (double)C);
    hipLaunchKernelGGL(gather_argmax_rows_kernel, dim3((unsigned)B), dim3(NT), 0, ctx.stream, ids, L, x, C, out);
    MAA_HIP(hipGetLastError());
}

void launch_l2norm_rows(const Ctx& ctx, const float* x, int B, int C, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "l2norm_rows_kernel", 0.0, 8.0 * B * (double)C);
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)B), dim3(NT), 0, ctx.stream, x, C, out);
    MAA_HIP(hipGetLastError());
}

void launch_gelu(const Ctx& ctx, const float* x, long long n, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "gelu_kernel", 0.0, 8.0 * (double)n);
    const long long blocks = (n + NT - 1) / NT;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(NT), 0, ctx.stream, x, n, out);
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
