// Small kernels of the waveform front ends and of the CLAP audio tower (clap_audio.cpp): everything there that is not a
// GEMM.  The contractions (framed DFT, mel filter bank, polyphase resampler, Cnn14's convolutions) run on the igemm engines.
#include "maa_internal.h"

namespace maa {

namespace {

constexpr int NT = 256;

// centre padding of librosa.stft / torchlibrosa.Spectrogram: x [B, n] -> out [B, ldo], out[b, i] = x[b, i - left] inside,
// mode 0: zeros outside (librosa >= 0.10 default, and the resampler's F.pad), mode 1: numpy "reflect" (no edge repeat)
__global__ __launch_bounds__(NT) void pad1d_kernel(const float* __restrict__ x, int B, int n, int left, int total, int ldo,
                                                    int mode, float* __restrict__ out) {
    const long long N = (long long)B * ldo;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < N; i += (long long)gridDim.x * NT) {
        const int b = (int)(i / ldo), j = (int)(i - (long long)b * ldo);
        int s = j - left;
        float v = 0.f;
        if (j < total) {
            if (mode == 1) {
                if (s < 0) s = -s;
                if (s >= n) s = 2 * (n - 1) - s;
            }
            if (s >= 0 && s < n) v = x[(long long)b * n + s];
        }
        out[i] = v;
    }
}

// y [rows, ldy] = (re | im) halves of nf columns each -> mag [rows, ldm]: re^2 + im^2 (power 2) or its square root
// (power 1: np.abs of librosa's complex STFT); columns nf .. ldm-1 are written as zeros (K padding of the mel GEMM)
__global__ __launch_bounds__(NT) void spec_power_kernel(const float* __restrict__ y, long long rows, int nf, int ldy, int ldm,
                                                         int power2, float* __restrict__ mag) {
    const long long N = rows * ldm;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < N; i += (long long)gridDim.x * NT) {
        const long long r = i / ldm;
        const int k = (int)(i - r * ldm);
        float v = 0.f;
        if (k < nf) {
            const float re = y[r * ldy + k], im = y[r * ldy + nf + k];
            v = re * re + im * im;
            if (!power2) v = sqrtf(v);
        }
        mag[i] = v;
    }
}

// mel [B * frames, ldmel] -> log-mel.
//   kind 0: torchlibrosa LogmelFilterBank.power_to_db with top_db = None: 10 log10(clamp(x, amin)) - 10 log10(max(amin, ref))
//   kind 1: TRANSFORMS_16000 after the mel product (extract_mel_spectrogram.py:140-150):
//           clip((20 log10(max(amin, x)) - 20 + 100) / 100, 0, 1)
//   layout 0: out [B, frames, n_mels];  layout 1: out [B, n_mels, frames] (the latent-diffusion model's mel image)
__global__ __launch_bounds__(NT) void logmel_kernel(const float* __restrict__ mel, int B, int frames, int n_mels, int ldmel,
                                                     int kind, float amin, float ref_db, int layout, float* __restrict__ out) {
    const long long N = (long long)B * frames * n_mels;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < N; i += (long long)gridDim.x * NT) {
        int b, t, m;
        if (layout == 0) {
            m = (int)(i % n_mels);
            const long long r = i / n_mels;
            t = (int)(r % frames);
            b = (int)(r / frames);
        } else {
            t = (int)(i % frames);
            const long long r = i / frames;
            m = (int)(r % n_mels);
            b = (int)(r / n_mels);
        }
        const float v = fmaxf(mel[((long long)b * frames + t) * ldmel + m], amin);
        float o;
        if (kind == 0) {
            o = 10.0f * log10f(v) - ref_db;
        } else {
            o = (log10f(v) * 20.f - 20.f + 100.f) / 100.f;
            o = fminf(fmaxf(o, 0.f), 1.f);
        }
        out[i] = o;
    }
}

// Cnn14.bn0 (CLAP/audio.py:150-152): BatchNorm2d over the mel axis of [B, 1, T, F]: y = x * scale[f] + shift[f]
__global__ __launch_bounds__(NT) void affine_lastdim_kernel(const float* __restrict__ x, long long n, int F,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int f = (int)(i % F);
        out[i] = x[i] * scale[f] + shift[f];
    }
}

// Cnn14 global pooling (audio.py:166-170) on channels-last x [B, T, F, C]: m[t] = mean_f x[b, t, f, c];
// out[b, c] = max_t m[t] + mean_t m[t].  One thread per (b, c): consecutive threads read consecutive channels.
__global__ __launch_bounds__(NT) void cnn14_pool_kernel(const float* __restrict__ x, int B, int T, int F, int C,
                                                         float* __restrict__ out) {
    const long long N = (long long)B * C;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < N; i += (long long)gridDim.x * NT) {
        const int b = (int)(i / C), c = (int)(i - (long long)b * C);
        float mx = -3.4e38f, sum = 0.f;
        for (int t = 0; t < T; ++t) {
            float s = 0.f;
            for (int f = 0; f < F; ++f) s += x[(((long long)b * T + t) * F + f) * C + c];
            const float m = s / (float)F;
            mx = fmaxf(mx, m);
            sum += m;
        }
        out[i] = mx + sum / (float)T;
    }
}

// CLAPWrapper.compute_similarity (CLAPWrapper.py:207-215): out[i, j] = scale * <audio[i], text[j]>; one wave per pair
__global__ __launch_bounds__(64) void similarity_kernel(const float* __restrict__ a, const float* __restrict__ t, int Nt, int D,
                                                         float scale, float* __restrict__ out) {
    const int i = blockIdx.x / Nt, j = blockIdx.x - i * Nt;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 64) s += a[(long long)i * D + d] * t[(long long)j * D + d];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = scale * s;
}

inline dim3 grid_for(long long n) {
    long long b = (n + NT - 1) / NT;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

}  // namespace

void launch_pad1d(const Ctx& ctx, const float* x, int B, int n, int left, int total, int ldo, int mode, float* out) {
    if (ctx.ws.dry) return;
    MAA_CHECK(mode == 0 || (n > left && n > total - left - n), "reflect padding needs a signal longer than the pad");
    ProfScope prof(ctx, "pad1d_kernel", 0.0, 8.0 * B * (double)ldo);
    hipLaunchKernelGGL(pad1d_kernel, grid_for((long long)B * ldo), dim3(NT), 0, ctx.stream, x, B, n, left, total, ldo, mode, out);
    MAA_HIP(hipGetLastError());
}

void launch_spec_power(const Ctx& ctx, const float* y, long long rows, int nf, int ldy, int ldm, int power2, float* mag) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "spec_power_kernel", 0.0, 4.0 * rows * (2.0 * nf + ldm));
    hipLaunchKernelGGL(spec_power_kernel, grid_for(rows * ldm), dim3(NT), 0, ctx.stream, y, rows, nf, ldy, ldm, power2, mag);
    MAA_HIP(hipGetLastError());
}

void launch_logmel(const Ctx& ctx, const float* mel, int B, int frames, int n_mels, int ldmel, int kind, float amin,
                   float ref_db, int layout, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "logmel_kernel", 0.0, 8.0 * B * (double)frames * n_mels);
    hipLaunchKernelGGL(logmel_kernel, grid_for((long long)B * frames * n_mels), dim3(NT), 0, ctx.stream, mel, B, frames, n_mels,
                       ldmel, kind, amin, ref_db, layout, out);
    MAA_HIP(hipGetLastError());
}

void launch_affine_lastdim(const Ctx& ctx, const float* x, long long n, int F, const float* scale, const float* shift,
                           float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "affine_lastdim_kernel", 0.0, 8.0 * (double)n);
    hipLaunchKernelGGL(affine_lastdim_kernel, grid_for(n), dim3(NT), 0, ctx.stream, x, n, F, scale, shift, out);
    MAA_HIP(hipGetLastError());
}

void launch_cnn14_pool(const Ctx& ctx, const float* x, int B, int T, int F, int C, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "cnn14_pool_kernel", 0.0, 4.0 * B * (double)T * F * C);
    hipLaunchKernelGGL(cnn14_pool_kernel, grid_for((long long)B * C), dim3(NT), 0, ctx.stream, x, B, T, F, C, out);
    MAA_HIP(hipGetLastError());
}

void launch_similarity(const Ctx& ctx, const float* audio, const float* text, int Na, int Nt, int D, float scale, float* out) {
    if (ctx.ws.dry) return;
    ProfScope prof(ctx, "similarity_kernel", 2.0 * Na * Nt * D, 4.0 * (Na + Nt) * (double)D);
    hipLaunchKernelGGL(similarity_kernel, dim3((unsigned)(Na * Nt)), dim3(64), 0, ctx.stream, audio, text, Nt, D, scale, out);
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
