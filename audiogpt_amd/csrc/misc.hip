// Small elementwise / layout kernels of the hot path (all HBM-bound, grid-stride, coalesced).
//   timestep embedding [cos | sin]      ldm/modules/diffusionmodules/util.py:151-171
//   CFG combine + DDIM x_{t-1} update   ldm/models/diffusion/ddim.py:199, 210-225
//   avg-pool 2x2 / nearest 2x           openaimodel.py:209-216 (resblock_updown), :116-118
//   BigVGAN Activation1d                vocoder/bigvgan/alias_free_torch/{act.py:23-27,resample.py:25-33,46-49,
//                                       filter.py:28-57,86-94}, activations.py:107-119
#include "maa_internal.h"

#include <mutex>

#include <cstdlib>

#include <cmath>

namespace maa {
namespace {

inline dim3 grid_for(long long n, int per_block = 256) {
    long long b = (n + per_block - 1) / per_block;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int B, int dim, float* __restrict__ out) {
    const int half = dim / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * half; i += gridDim.x * blockDim.x) {
        const int b = i / half, j = i - b * half;
        // freqs = exp(-ln(10000) * j / half) in fp32, as torch.exp over an fp32 arange
        const float f = expf(-9.210340371976184f * (float)j / (float)half);
        const float a = t[b] * f;
        out[(long long)b * dim + j] = cosf(a);
        out[(long long)b * dim + half + j] = sinf(a);
    }
}

__global__ void silu_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        out[i] = v / (1.f + expf(-v));
    }
}

__global__ void leaky_kernel(const float* __restrict__ x, long long n, float slope, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        out[i] = v > 0.f ? v : v * slope;
    }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                           float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

__global__ void scale_kernel(const float* __restrict__ x, long long n, float s, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = x[i] * s;
}

__global__ void clamp_affine_kernel(const float* __restrict__ x, long long n, float mul, float add, float lo,
                                    float hi, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(x[i] * mul + add, lo), hi);
}

// x [B, C, HW] -> out [B, HW, C]
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int B, int C, int HW, float* __restrict__ out) {
    const long long n = (long long)B * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long r = i / C;
        const int p = (int)(r % HW);
        const int b = (int)(r / HW);
        out[i] = x[((long long)b * C + c) * HW + p];
    }
}

// x [B, HW, ld_in] (first C channels) -> out [B, C, HW]
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int B, int C, int HW, int ld_in,
                                    float* __restrict__ out) {
    const long long n = (long long)B * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long long r = i / HW;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        out[i] = x[((long long)b * HW + p) * ld_in + c];
    }
}

// clamp((mel + 1) / 2, 0, 1): the line between decode_first_stage and the vocoder in every tool (audio-chatgpt.py:175-176,
// 254-255, 521-522), in the reference's operation order
__global__ void spec_from_mel_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf((x[i] + 1.0f) / 2.0f, 0.0f), 1.0f);
}

// 3x3 convolution (stride 1, padding 1) to a FEW output channels (N <= 4: the UNet's 320 -> 4 output convolution,
// openaimodel.py:693-697) from split32 rows, written straight to NCHW.  The implicit-GEMM engines pad N to 32 and ran this layer
// on 49 - 98 workgroups of 90 K chunks (58 us for 0.3 GFLOP); here a wave owns four consecutive output positions, a lane eight
// channels (one 16-byte load of the hi halves and one of the lo halves per tap and position; x = hi + lo, 2^-17 relative), the
// weights sit in LDS as one float4 per (tap, channel) and are read once per four positions, the products are fp32 FMAs (more
// exact than the bf16x3 MFMA), the 16 partial sums are reduced across the wave at the end.
__global__ __launch_bounds__(256) void narrow_conv3x3_kernel(const float* __restrict__ a, int lda, int B, int H, int W, int C,
                                                            const float4* __restrict__ w4, const float* __restrict__ bias, int N,
                                                            float* __restrict__ out_nchw) {
    extern __shared__ __attribute__((aligned(16))) float4 sw[];      // [9 taps][8 k][C / 8 groups]: channel 8 g + k of tap t
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 9 * C; i += 256) sw[i] = w4[i];
    __syncthreads();
    const int HW = H * W;
    const long long M = (long long)B * HW;
    const long long m0 = ((long long)blockIdx.x * 4 + wid) * 4;
    if (m0 >= M) return;
    int py[4], px[4];
    bool live[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long long m = m0 + p;
        live[p] = m < M;
        const int r = (int)((live[p] ? m : 0) % HW);
        py[p] = r / W;
        px[p] = r - py[p] * W;
    }
    float acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[p][n] = 0.f;
    const char* ab = reinterpret_cast<const char*>(a);
    // (measured, round 6: fetching the next tap's pieces before multiplying this one costs 70 more VGPRs and doubles the kernel's
    // time; unrolling the taps spills)
    for (int g = lane; g < C / 8; g += 64) {
        const long long goff = (long long)(g >> 2) * 128 + (g & 3) * 16;      // hi halves of channels 8 g .. 8 g + 7 inside a row
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            float x[4][8];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int iy = py[p] + dy, ix = px[p] + dx;
                const bool ok = live[p] && iy >= 0 && iy < H && ix >= 0 && ix < W;      // wave-uniform
                if (ok) {
                    const char* row = ab + ((m0 + p) + (long long)dy * W + dx) * (long long)lda * 4 + goff;
                    const uint4 hi = *reinterpret_cast<const uint4*>(row);
                    const uint4 lo = *reinterpret_cast<const uint4*>(row + 64);
                    const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        x[p][2 * k] = __builtin_bit_cast(float, hw[k] << 16) + __builtin_bit_cast(float, lw[k] << 16);
                        x[p][2 * k + 1] = __builtin_bit_cast(float, hw[k] & 0xffff0000u) + __builtin_bit_cast(float, lw[k] & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) x[p][k] = 0.f;
                }
            }
            const int G = C >> 3;
            const float4* wt = sw + (t * 8) * G + g;      // [tap][k][g]: the lanes of a wave read consecutive float4 (no bank conflicts)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 wv = wt[k * G];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    acc[p][0] = fmaf(x[p][k], wv.x, acc[p][0]);
                    acc[p][1] = fmaf(x[p][k], wv.y, acc[p][1]);
                    acc[p][2] = fmaf(x[p][k], wv.z, acc[p][2]);
                    acc[p][3] = fmaf(x[p][k], wv.w, acc[p][3]);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v = acc[p][n];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[p][n] = v;
        }
    if (lane < 16) {
        const int p = lane >> 2, n = lane & 3;
        const long long m = m0 + p;
        if (m < M && n < N) {
            float v = 0.f;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
                    if (pp == p && nn == n) v = acc[pp][nn];
            const long long b = m / HW;
            out_nchw[(b * N + n) * HW + (m - b * HW)] = v + bias[n];
        }
    }
}

// planes [4 = (py, px)][B, H, W, C] -> out [B, 2H, 2W, C]: the four phase outputs of the up2 convolution interleaved into the image
// (16 bytes per thread; a pixel's C channels are one contiguous run on both sides)
__global__ void pixel_shuffle2_kernel(const float4* __restrict__ planes, int B, int H, int W, int C4, float4* __restrict__ out) {
    const long long n = (long long)B * H * W * 4 * C4;
    const long long plane = (long long)B * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % (2 * W));
        r /= 2 * W;
        const int oy = (int)(r % (2 * H));
        const int b = (int)(r / (2 * H));
        const int phase = (oy & 1) * 2 + (ox & 1);
        out[i] = planes[phase * plane + (((long long)b * H + (oy >> 1)) * W + (ox >> 1)) * C4 + c];
    }
}

__global__ void avgpool2_kernel(const float* __restrict__ x, int B, int H, int W, int C, float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const long long n = (long long)B * Ho * Wo * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float* s = x + (((long long)b * H + 2 * oy) * W + 2 * ox) * C + c;
        // same summation order as ATen's avg_pool2d (row-major over the window), then divide
        out[i] = (s[0] + s[C] + s[(long long)W * C] + s[(long long)W * C + C]) / 4.f;
    }
}

__global__ void upsample2_kernel(const float* __restrict__ x, int B, int H, int W, int C, float* __restrict__ out) {
    const int Ho = H * 2, Wo = W * 2;
    const long long n = (long long)B * Ho * Wo * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        out[i] = x[(((long long)b * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
    }
}

// (x and x_prev may be the same buffer -- the sampler updates the latent in place -- so neither is __restrict__)
__global__ void ddim_update_kernel(const float* x, const float* __restrict__ eu, const float* __restrict__ ec, float scale,
                                   const float* __restrict__ coef, long long n, float* x_prev,
                                   float* __restrict__ pred_x0, int* __restrict__ step) {
    if (step && blockIdx.x == 0 && threadIdx.x == 0) *step = *step - 1;      // next DDIM index (read by the next prepare)
    const float a_t = coef[0], a_prev = coef[1], sigma = coef[2], somat = coef[3];
    const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev), dir = sqrtf(1.f - a_prev - sigma * sigma);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float e = eu[i];
        if (ec) e = e + scale * (ec[i] - e);
        const float x0 = (x[i] - somat * e) / sqrt_at;
        x_prev[i] = sqrt_ap * x0 + dir * e;
        if (pred_x0) pred_x0[i] = x0;
    }
}

// The sampling loop's form of the update (ddim.py:199, 210-225 in full): x is read from the step's UNet input (the first B
// samples of xin hold the -- possibly mask-blended -- latent, per_in elements apart), noise = sigma_t * z * temperature with the
// caller's z of this step (eta > 0), and the step's x_{t-1} / pred_x0 are copied into the log slabs when the device table
// says this index is logged (ddim.py:161-163).  Term order as the reference: (sqrt(a_prev) x0 + dir e) + noise.
__global__ void ddim_step_kernel(const float* __restrict__ xin, long long per, long long per_in, const float* __restrict__ eu,
                                 const float* __restrict__ ec, float scale, const float* __restrict__ coef, long long n,
                                 float* __restrict__ x_prev, const float* __restrict__ noise_p, float temperature, int S,
                                 float* __restrict__ log_x, float* __restrict__ log_x0, int* __restrict__ step) {
    // (the DDIM index comes from the step's coefficient slot, written by the prepare kernel: *step itself is decremented
    //  below by one thread while other blocks may still be starting)
    const int idx = (int)coef[7];
    if (blockIdx.x == 0 && threadIdx.x == 0) *step = idx - 1;
    const float a_t = coef[0], a_prev = coef[1], sigma = coef[2], somat = coef[3];
    const int slot = (int)coef[6];
    const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev), dir = sqrtf(1.f - a_prev - sigma * sigma);
    const float* z = noise_p ? noise_p + (long long)(S - 1 - idx) * n : nullptr;      // visiting order: i = S - 1 - index
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float e = eu[i];
        if (ec) e = e + scale * (ec[i] - e);
        const long long b = i / per;
        const float xv = xin[b * per_in + (i - b * per)];
        const float x0 = (xv - somat * e) / sqrt_at;
        float xp = sqrt_ap * x0 + dir * e;
        if (z) xp = xp + sigma * z[i] * temperature;
        x_prev[i] = xp;
        if (slot >= 0 && log_x) {
            log_x[(long long)slot * n + i] = xp;
            log_x0[(long long)slot * n + i] = x0;
        }
    }
}

// UNet input and scalars of one DDIM step, with nothing from the host: idx = *step selects the row of the device
// tables; xin[b'] = cat(x[b' % B], concat[b' % B]) for b' < nB (nB = 2B duplicates the latents for CFG in the order
// [uncond ; cond], ddim.py:177-179; concat is the inpaint model's conditioning, ddpm.py:1404-1406).
// mask != null (ddim.py:147-150): the latent is first blended with the noised original,
//   img = q_sample(x0, t) * mask + (1 - mask) * img,   q_sample = sqrt(ac_t) x0 + sqrt(1 - ac_t) z   (ddpm.py:272-275)
// with the caller's z of this step; the blended latent only exists inside xin (the update kernel reads it from there).
__global__ void ddim_prepare_kernel(const float* __restrict__ x, const float* __restrict__ concat, int B, int nB,
                                    long long per, long long per_c, const float* __restrict__ tab_t,
                                    const float* __restrict__ tab_coef, const int* __restrict__ step,
                                    float* __restrict__ xin, float* __restrict__ cur_t, float* __restrict__ cur_coef,
                                    const float* __restrict__ mask, const float* __restrict__ x0, const float* __restrict__ noise_q,
                                    int S, const float* __restrict__ emb_tab, int emb_w, float* __restrict__ cur_emb) {
    const int idx = *step;
    if (emb_tab)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < emb_w; i += gridDim.x * blockDim.x) cur_emb[i] = emb_tab[(long long)idx * emb_w + i];
    if (blockIdx.x == 0) {
        if (threadIdx.x < nB) cur_t[threadIdx.x] = tab_t[idx];
        if (threadIdx.x < 8) cur_coef[threadIdx.x] = tab_coef[idx * 8 + threadIdx.x];
    }
    const float sq_ac = tab_coef[idx * 8 + 4], sq_1mac = tab_coef[idx * 8 + 5];
    const float* z = noise_q ? noise_q + (long long)(S - 1 - idx) * B * per : nullptr;
    const long long per_in = per + per_c, n = (long long)nB * per_in;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int bb = (int)(i / per_in);
        const long long r = i - bb * per_in;
        const int b = bb % B;
        float v;
        if (r < per) {
            v = x[b * per + r];
            if (mask) {
                const long long e = b * per + r;
                const float orig = sq_ac * x0[e] + sq_1mac * z[e];
                v = orig * mask[e] + (1.f - mask[e]) * v;
            }
        } else {
            v = concat[b * per_c + (r - per)];
        }
        xin[i] = v;
    }
}

// 12-tap Kaiser-sinc low-pass (cutoff 0.25, half-width 0.3): alias_free_torch/filter.py:28-57 restated on the host
__constant__ float c_fir12[12];

static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 64; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

static void kaiser_sinc_filter12(float* out) {
    const int ks = 12, half = 6;
    const double cutoff = 0.25, half_width = 0.3, pi = 3.14159265358979323846;
    const double delta_f = 4.0 * half_width;
    const double A = 2.285 * (half - 1) * pi * delta_f + 7.95;
    double beta = 0.0;
    if (A > 50.0)
        beta = 0.1102 * (A - 8.7);
    else if (A >= 21.0)
        beta = 0.5842 * std::pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0);
    double f[12], sum = 0.0;
    for (int n = 0; n < ks; ++n) {
        const double r = (2.0 * n) / (ks - 1) - 1.0;                       // kaiser_window(periodic=False)
        const double w = bessel_i0(beta * std::sqrt(1.0 - r * r)) / bessel_i0(beta);
        const double t = (n - half) + 0.5;                                 // even kernel: arange(-6, 6) + 0.5
        const double xx = 2.0 * cutoff * t;
        const double sinc = xx == 0.0 ? 1.0 : std::sin(pi * xx) / (pi * xx);
        f[n] = 2.0 * cutoff * w * sinc;
        sum += f[n];
    }
    for (int n = 0; n < ks; ++n) out[n] = (float)(f[n] / sum);
}

// out[b,l,c] = sum_j f[j] * snake( up[2l + j - 5] ),  up[u] = 2 * sum_i f[.] x_rep[...]   (replicate edges)
//   UpSample1d : pad 5 replicate -> 2*conv_transpose(stride 2) -> crop [15:-15]       resample.py:25-33
//   DownSample1d: pad (5,6) replicate -> conv stride 2                                 filter.py:86-94
__global__ void snake_aa_kernel(const float* __restrict__ x, int B, int L, int C, const float* __restrict__ inv_beta,
                                const float* __restrict__ alpha, float* __restrict__ out) {
    const long long n = (long long)B * L * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long r = i / C;
        const int l = (int)(r % L);
        const int b = (int)(r / L);
        const float* xs = x + (long long)b * L * C + c;
        const float al = alpha[c], ib = inv_beta[c];
        const int U = 2 * L;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            int u = 2 * l + j - 5;                 // index into the upsampled signal, replicate-clamped
            u = u < 0 ? 0 : (u >= U ? U - 1 : u);
            // upsampled[u] = 2 * sum_k f[k] * xpad[(u + 15 - k) / 2] for (u + 15 - k) even, xpad = replicate pad 5
            float up = 0.f;
            const int par = (u + 15) & 1;
#pragma unroll
            for (int k2 = 0; k2 < 6; ++k2) {
                const int k = 2 * k2 + par;
                int src = ((u + 15 - k) >> 1) - 5;
                src = src < 0 ? 0 : (src >= L ? L - 1 : src);
                up += c_fir12[k] * xs[(long long)src * C];
            }
            up *= 2.f;
            const float sn = sinf(up * al);
            acc += c_fir12[j] * (up + ib * sn * sn);
        }
        out[i] = acc;
    }
}

// The same Activation1d with every up-sampled, snake-activated value computed ONCE per tile instead of once per output
// tap that reads it (each is read by six outputs): a block owns 64 positions x CC channels, stages the x rows it needs
// (l0 - 6 .. l0 + 68, replicate-clamped) in LDS, forms snake(up[u]) for u = 2 l0 - 5 .. 2 l0 + 132 in LDS, then each thread
// applies the 12-tap down filter.  Per output: 2.2 sinf and ~25 FMAs instead of 12 sinf and ~84.  Arithmetic and summation
// order are those of snake_aa_kernel (bit-identical results).
template <int CC>
__global__ __launch_bounds__(256) void snake_aa_tiled_kernel(const float* __restrict__ x, int L, int C,
                                                             const float* __restrict__ inv_beta,
                                                             const float* __restrict__ alpha, float* __restrict__ out) {
    constexpr int TLO = 64, XR = TLO + 11, NU = 2 * TLO + 10, LANES = 256 / CC;
    __shared__ float xs[XR][CC];
    __shared__ float us[NU][CC];
    const int c = threadIdx.x % CC, s = threadIdx.x / CC;
    const int c0 = blockIdx.y * CC, l0 = blockIdx.x * TLO, b = blockIdx.z;
    const float* xb = x + (long long)b * L * C + c0 + c;
    for (int r = s; r < XR; r += LANES) {
        int src = l0 - 6 + r;
        src = src < 0 ? 0 : (src >= L ? L - 1 : src);
        xs[r][c] = xb[(long long)src * C];
    }
    __syncthreads();
    const float al = alpha[c0 + c], ib = inv_beta[c0 + c];
    const int U = 2 * L;
    for (int idx = s; idx < NU; idx += LANES) {
        int u = 2 * l0 - 5 + idx;
        u = u < 0 ? 0 : (u >= U ? U - 1 : u);
        float up = 0.f;
        const int par = (u + 15) & 1;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
            const int k = 2 * k2 + par;
            int src = ((u + 15 - k) >> 1) - 5;
            src = src < 0 ? 0 : (src >= L ? L - 1 : src);
            up += c_fir12[k] * xs[src - (l0 - 6)][c];
        }
        up *= 2.f;
        const float sn = sinf(up * al);
        us[idx][c] = up + ib * sn * sn;
    }
    __syncthreads();
    for (int q = s; q < TLO; q += LANES) {
        const int l = l0 + q;
        if (l < L) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) acc += c_fir12[j] * us[2 * q + j][c];
            out[((long long)b * L + l) * C + c0 + c] = acc;
        }
    }
}

// the two halves of a guided DDIM step's batch, cat([x] * 2), are one tensor until the first cross-attention (unet.cpp): the
// layers before it run on one half and this copy makes the [2B] tensor the rest of the network reads
__global__ void dup_half_kernel(const float4* __restrict__ x, long long n4, float4* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        out[i] = v;
        out[i + n4] = v;
    }
}

}  // namespace

#define MAA_LAUNCH1(kern, n, ...)                                                            \
    if (ctx.ws.dry) return;                                                                  \
    ProfScope prof(ctx, #kern, 0.0, 8.0 * (double)(n));                                      \
    hipLaunchKernelGGL(kern, grid_for(n), dim3(256), 0, ctx.stream, __VA_ARGS__);            \
    MAA_HIP(hipGetLastError())

void launch_timestep_embedding(const Ctx& ctx, const float* t, int B, int dim, float* out) {
    MAA_LAUNCH1(timestep_embedding_kernel, (long long)B * dim / 2, t, B, dim, out);
}
void launch_silu(const Ctx& ctx, const float* x, long long n, float* out) { MAA_LAUNCH1(silu_kernel, n, x, n, out); }
void launch_leaky(const Ctx& ctx, const float* x, long long n, float slope, float* out) {
    MAA_LAUNCH1(leaky_kernel, n, x, n, slope, out);
}
void launch_add(const Ctx& ctx, const float* a, const float* b, long long n, float* out) {
    MAA_LAUNCH1(add_kernel, n, a, b, n, out);
}
void launch_scale(const Ctx& ctx, const float* x, long long n, float s, float* out) {
    MAA_LAUNCH1(scale_kernel, n, x, n, s, out);
}
void launch_clamp_affine(const Ctx& ctx, const float* x, long long n, float mul, float add, float lo, float hi,
                         float* out) {
    MAA_LAUNCH1(clamp_affine_kernel, n, x, n, mul, add, lo, hi, out);
}
void launch_spec_from_mel(const Ctx& ctx, const float* mel, long long n, float* spec) {
    MAA_LAUNCH1(spec_from_mel_kernel, n, mel, n, spec);
}
void launch_nchw_to_nhwc(const Ctx& ctx, const float* x, int B, int C, int HW, float* out) {
    MAA_LAUNCH1(nchw_to_nhwc_kernel, (long long)B * C * HW, x, B, C, HW, out);
}
void launch_nhwc_to_nchw(const Ctx& ctx, const float* x, int B, int C, int HW, float* out, int ld_in) {
    MAA_LAUNCH1(nhwc_to_nchw_kernel, (long long)B * C * HW, x, B, C, HW, ld_in, out);
}
bool launch_narrow_conv3x3(const Ctx& ctx, const float* a_split, int lda, int B, int H, int W, int C, const float* w4, const float* bias,
                           int N, float* out_nchw) {
    const size_t lds = (size_t)9 * C * sizeof(float4);
    if (ctx.dtype != 1 || N > 4 || C % 32 != 0 || lds > 98304 || !w4 || !bias) return false;
    if (ctx.ws.dry) return true;
    const long long M = (long long)B * H * W;
    ProfScope prof(ctx, "narrow_conv3x3_kernel", 2.0 * M * N * 9.0 * C, 4.0 * M * C);
    ensure_dynamic_lds(reinterpret_cast<const void*>(narrow_conv3x3_kernel), ctx.device, (int)lds);
    hipLaunchKernelGGL(narrow_conv3x3_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), lds, ctx.stream, a_split, lda, B, H, W, C,
                       reinterpret_cast<const float4*>(w4), bias, N, out_nchw);
    MAA_HIP(hipGetLastError());
    return true;
}
void launch_pixel_shuffle2(const Ctx& ctx, const float* planes, int B, int H, int W, int C, float* out) {
    MAA_CHECK(C % 4 == 0, "pixel_shuffle2: channels must be a multiple of 4");
    MAA_LAUNCH1(pixel_shuffle2_kernel, (long long)B * H * W * C, reinterpret_cast<const float4*>(planes), B, H, W, C / 4,
                reinterpret_cast<float4*>(out));
}
void launch_dup_half(const Ctx& ctx, const float* x, long long n, float* out) {
    MAA_CHECK(n % 4 == 0, "dup_half: element count must be a multiple of 4");
    MAA_LAUNCH1(dup_half_kernel, n / 4, reinterpret_cast<const float4*>(x), n / 4, reinterpret_cast<float4*>(out));
}
void launch_avgpool2(const Ctx& ctx, const float* x, int B, int H, int W, int C, float* out) {
    MAA_LAUNCH1(avgpool2_kernel, (long long)B * (H / 2) * (W / 2) * C, x, B, H, W, C, out);
}
void launch_upsample2(const Ctx& ctx, const float* x, int B, int H, int W, int C, float* out) {
    MAA_LAUNCH1(upsample2_kernel, (long long)B * H * W * C * 4, x, B, H, W, C, out);
}
void launch_ddim_update(const Ctx& ctx, const float* x, const float* eps_u, const float* eps_c, float scale,
                        const float* coef, long long n, float* x_prev, float* pred_x0, int* step) {
    MAA_LAUNCH1(ddim_update_kernel, n, x, eps_u, eps_c, scale, coef, n, x_prev, pred_x0, step);
}
void launch_ddim_prepare(const Ctx& ctx, const float* x, const float* concat, int B, int nB, long long per,
                         long long per_c, const float* tab_t, const float* tab_coef, const int* step, float* xin,
                         float* cur_t, float* cur_coef, const float* mask, const float* x0, const float* noise_q, int S,
                         const float* emb_tab, int emb_w, float* cur_emb) {
    MAA_CHECK(nB <= 256, "ddim: at most 256 UNet rows per step");
    MAA_LAUNCH1(ddim_prepare_kernel, (long long)nB * (per + per_c), x, concat, B, nB, per, per_c, tab_t, tab_coef, step,
                xin, cur_t, cur_coef, mask, x0, noise_q, S, emb_tab, emb_w, cur_emb);
}
void launch_ddim_step(const Ctx& ctx, const float* xin, long long per, long long per_in, const float* eps_u, const float* eps_c,
                      float scale, const float* coef, long long n, float* x_prev, const float* noise_p, float temperature, int S,
                      float* log_x, float* log_x0, int* step) {
    MAA_LAUNCH1(ddim_step_kernel, n, xin, per, per_in, eps_u, eps_c, scale, coef, n, x_prev, noise_p, temperature, S, log_x, log_x0,
                step);
}

static std::once_flag g_fir_once[64];      // one upload of the FIR taps per device (thread-safe: contexts on several threads)
void launch_snake_aa(const Ctx& ctx, const float* x, int B, int L, int C, const float* inv_beta, const float* alpha,
                     float* out) {
    if (ctx.ws.dry) return;
    MAA_CHECK(ctx.device >= 0 && ctx.device < 64, "device index");
    std::call_once(g_fir_once[ctx.device], [] {
        float f[12];
        kaiser_sinc_filter12(f);
        MAA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_fir12), f, sizeof(f)));
    });
    if ((C % 64 == 0 || C == 32) && B <= 65535) {
        ProfScope prof(ctx, "snake_aa_kernel", 0.0, 8.0 * (double)B * L * C);
        const dim3 grid((unsigned)((L + 63) / 64), (unsigned)(C % 64 == 0 ? C / 64 : 1), (unsigned)B);
        if (C % 64 == 0)
            hipLaunchKernelGGL(snake_aa_tiled_kernel<64>, grid, dim3(256), 0, ctx.stream, x, L, C, inv_beta, alpha, out);
        else
            hipLaunchKernelGGL(snake_aa_tiled_kernel<32>, grid, dim3(256), 0, ctx.stream, x, L, C, inv_beta, alpha, out);
        MAA_HIP(hipGetLastError());
        return;
    }
    MAA_LAUNCH1(snake_aa_kernel, (long long)B * L * C, x, B, L, C, inv_beta, alpha, out);
}

}  // namespace maa
