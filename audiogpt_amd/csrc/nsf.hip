// Harmonic-plus-noise source of the NSF (f0-conditioned) HiFi-GAN branch.
//
// Replaces, for the singing tools' vocoder (NeuralSeq/modules/hifigan/hifigan.py:111-115, 145-149):
//   f0_upsamp (nearest, x hop)  ->  SineGen.forward (NeuralSeq/modules/parallel_wavegan/models/source.py:399-436, with
//   _f02sine :346-397 in its non-flag_for_pulse branch)  ->  SourceModuleHnNSF: tanh(Linear(9 -> 1)) (:526-535)
// The reference draws two random tensors inside SineGen.forward (torch.rand for the initial phase of the overtones,
// torch.randn_like for the additive noise); here they are INPUTS (rand_ini [B, H+1], noise [B, L, H+1]), so that a
// caller reproducing the reference's draws gets the reference's waveform.
//
// Numerics.  The phase is a cumulative sum over up to ~10^5 samples.  ATen's CPU cumsum accumulates fp32 data in double
// and rounds each prefix to fp32; so does this kernel (a per-thread chunk + block prefix in double: the summation order
// then only matters at the 1e-13 level, far below the fp32 rounding of each prefix).  Everything else (the % 1 wraps,
// the -1 phase shift where the wrapped sum steps down, sin, the noise mix) is elementwise fp32 as in the reference.
#include "maa_internal.h"

namespace maa {

namespace {

constexpr int NT = 256;

// one block per (harmonic h, sample b): out[b, l, h] for l in [0, L)
__global__ __launch_bounds__(NT) void nsf_sine_kernel(const float* __restrict__ f0, int T, int hop, float sampling_rate,
                                                     const float* __restrict__ rand_ini, const float* __restrict__ noise,
                                                     int H1, float sine_amp, float noise_std, float* __restrict__ out) {
    __shared__ double part[NT];
    __shared__ double base[NT + 1];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int L = T * hop;
    const int per = (L + NT - 1) / NT;
    const int l0 = min(tid * per, L), l1 = min(l0 + per, L);
    const float mult = (float)(h + 1);
    const float ini = h == 0 ? 0.f : rand_ini[b * H1 + h];          // the fundamental starts at phase 0 (source.py:357)
    const float* f0b = f0 + (long long)b * T;
    auto rad_at = [&](int l) {
        const float fb = f0b[l / hop] * mult;                         // f0_buf (:409-413), nearest-upsampled f0
        const float q = fb / sampling_rate;
        float r = q - floorf(q);                                      // % 1 (:349)
        if (l == 0) r = r + ini;                                      // (:358)
        return r;
    };
    auto block_prefix = [&](double local) {      // exclusive prefix of `local` over the block, in thread order
        part[tid] = local;
        __syncthreads();
        if (tid == 0) {
            double run = 0.0;
            for (int t = 0; t < NT; ++t) {
                base[t] = run;
                run += part[t];
            }
        }
        __syncthreads();
        const double r = base[tid];
        __syncthreads();
        return r;
    };
    // pass A: prefix of rad
    double s1 = 0.0;
    for (int l = l0; l < l1; ++l) s1 += (double)rad_at(l);
    const double p1 = block_prefix(s1);
    // wrapped cumulative sum one sample back (tmp_over_one, :369), needed for the first sample of the chunk
    auto wrap = [](double run) {
        const float t = (float)run;
        return t - floorf(t);
    };
    // pass B: prefix of rad + shift
    double run1 = p1, s2 = 0.0;
    float prev = wrap(run1);
    for (int l = l0; l < l1; ++l) {
        const float r = rad_at(l);
        run1 += (double)r;
        const float cur = wrap(run1);
        const float shift = (l > 0 && (cur - prev) < 0.f) ? -1.0f : 0.f;      // (:370-373)
        prev = cur;
        s2 += (double)(r + shift);
    }
    const double p2 = block_prefix(s2);
    // pass C: sines, voiced mask, noise mix
    run1 = p1;
    double run2 = p2;
    prev = wrap(run1);
    const float third = sine_amp / 3.0f;
    for (int l = l0; l < l1; ++l) {
        const float r = rad_at(l);
        run1 += (double)r;
        const float cur = wrap(run1);
        const float shift = (l > 0 && (cur - prev) < 0.f) ? -1.0f : 0.f;
        prev = cur;
        run2 += (double)(r + shift);
        const float ph = (float)run2 * 2.0f * 3.14159265358979323846f;         // cumsum(...) * 2 * np.pi (:375-376)
        const float sv = sinf(ph) * sine_amp;                                   // (:416)
        const float uv = f0b[l / hop] > 0.f ? 1.0f : 0.f;                       // (:340-344), voiced_threshold 0
        const float namp = uv * noise_std + (1.0f - uv) * third;                // (:424)
        const long long at = ((long long)b * L + l) * H1 + h;
        out[at] = sv * uv + namp * noise[at];                                   // (:425-429)
    }
}

// har[b, l] = tanh(sum_h w[h] * sines[b, l, h] + bias)      (source.py:533)
__global__ __launch_bounds__(NT) void nsf_merge_kernel(const float* __restrict__ sines, long long n, int H1,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ har) {
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float* s = sines + i * H1;
        float acc = 0.f;
        for (int h = 0; h < H1; ++h) acc += s[h] * w[h];
        har[i] = tanhf(acc + bias[0]);
    }
}

}  // namespace

void launch_nsf_source(const Ctx& ctx, const float* f0, int B, int T, int hop, float sampling_rate, const float* rand_ini,
                       const float* noise, int harmonics, const float* w, const float* bias, float* sines, float* har) {
    if (ctx.ws.dry) return;
    const int H1 = harmonics + 1;
    const long long n = (long long)B * T * hop;
    {
        ProfScope prof(ctx, "nsf_sine_kernel", 0.0, 8.0 * (double)n * H1);
        hipLaunchKernelGGL(nsf_sine_kernel, dim3((unsigned)H1, (unsigned)B), dim3(NT), 0, ctx.stream, f0, T, hop, sampling_rate,
                           rand_ini, noise, H1, 0.1f, 0.003f, sines);
    }
    {
        ProfScope prof(ctx, "nsf_merge_kernel", 0.0, 4.0 * (double)n * (H1 + 1));
        const long long blocks = (n + NT - 1) / NT;
        hipLaunchKernelGGL(nsf_merge_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(NT), 0, ctx.stream, sines, n,
                           H1, w, bias, har);
    }
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
