// Elementwise kernels of DiffSinger's denoiser (DiffNet) and of the PLMS sampling loop, channels-last [B, T, C].
//
// Replaces (NeuralSeq/modules/diff/): net.py:31-44 SinusoidalPosEmb, diffusion.py:68-70 Mish, net.py:68-81 the gated
// activation and the residual / skip bookkeeping of ResidualBlock, shallow_diffusion_tts.py:166-201 p_sample_plms
// (get_x_pred and the 1st..4th-order pseudo linear multistep combinations of the noise history).
#include "maa_internal.h"

namespace maa {

namespace {

inline dim3 grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

// [sin | cos] of t * exp(-ln(1e4) * j / (half - 1))      (net.py:36-44)
__global__ void ds_pos_emb_kernel(const float* __restrict__ t, int B, int dim, float* __restrict__ out) {
    const int half = dim / 2;
    const float scale = 9.210340371976184f / (float)(half - 1);          // math.log(10000) / (half_dim - 1)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * half; i += gridDim.x * blockDim.x) {
        const int b = i / half, j = i - b * half;
        const float f = expf((float)j * -scale);
        const float a = t[b] * f;
        out[(long long)b * dim + j] = sinf(a);
        out[(long long)b * dim + half + j] = cosf(a);
    }
}

// x * tanh(softplus(x)), softplus with torch's threshold 20 (F.softplus default)
__global__ void ds_mish_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        const float sp = v > 20.f ? v : log1pf(expf(v));
        out[i] = v * tanhf(sp);
    }
}

// out[b, t, c] = x[b, t, c] + step[b, c]       (net.py:71-73: y = x + diffusion_step, broadcast over time)
__global__ void ds_add_step_kernel(const float* __restrict__ x, const float* __restrict__ step, int ld_step, long long rows,
                                   int T, int C, float* __restrict__ out) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        out[i] = x[i] + step[(r / T) * ld_step + c];
    }
}

// gate, filter = chunk(y, 2, dim = channels); z = sigmoid(gate) * tanh(filter)      (net.py:76-77)
__global__ void ds_gate_kernel(const float* __restrict__ y, long long rows, int C, float* __restrict__ z) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float g = y[r * 2 * C + c], f = y[r * 2 * C + C + c];
        z[i] = (1.f / (1.f + expf(-g))) * tanhf(f);
    }
}

// residual, skip = chunk(y2, 2); x = (x + residual) / sqrt(2); skip_sum (+)= skip; xin = x + next_step[b]   (net.py:79-81)
__global__ void ds_residual_kernel(const float* __restrict__ y2, long long rows, int T, int C, float* __restrict__ x,
                                   float* __restrict__ skip, int first, const float* __restrict__ next_step, int ld_step,
                                   float* __restrict__ xin) {
    const long long n = rows * C;
    const float rs2 = 1.41421356237309504880f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float xn = (x[i] + y2[r * 2 * C + c]) / rs2;
        x[i] = xn;
        const float s = y2[r * 2 * C + C + c];
        skip[i] = first ? s : skip[i] + s;
        if (next_step) xin[i] = xn + next_step[(r / T) * ld_step + c];
    }
}

// One PLMS step (shallow_diffusion_tts.py:166-201).  st = {t index (counts down by `interval`), history count}.
// mode 0: e' = combination of e and the history by its count (1: (3e - h1)/2, 2: (23e - 16h1 + 5h2)/12,
//         >= 3: (55e - 59h1 + 37h2 - 9h3)/24); x = get_x_pred(x, e', t); history <- e; t -= interval
// mode 1: x_out = get_x_pred(x, e, t)                       (the predictor of the very first step)
// mode 2: e' = (e + e_prev) / 2; x = get_x_pred(x, e', t); history <- e; t -= interval     (its corrector)
// hist: ring of 3 buffers [3][n]; slot of h_k = (head - k) mod 3, head = st[2].
__global__ void ds_plms_kernel(float* x, const float* __restrict__ e, const float* __restrict__ e_prev, float* hist,
                               long long n, const float* __restrict__ ac, int interval, int* __restrict__ st, int mode,
                               float* x_out) {
    const int t = st[0], cnt = st[1], head = st[2];
    const float a_t = ac[t];
    const float a_prev = t < interval ? 1.0f : ac[max(t - interval, 0)];
    const float a_t_sq = sqrtf(a_t), a_prev_sq = sqrtf(a_prev);
    const float c1 = 1.0f / (a_t_sq * (a_t_sq + a_prev_sq));
    const float c2 = 1.0f / (a_t_sq * (sqrtf((1.0f - a_prev) * a_t) + sqrtf((1.0f - a_t) * a_prev)));
    const float da = a_prev - a_t;
    const float* h1 = hist + (long long)((head + 3) % 3) * n;
    const float* h2 = hist + (long long)((head + 2) % 3) * n;
    const float* h3 = hist + (long long)((head + 1) % 3) * n;
    float* hnew = hist + (long long)((head + 1) % 3) * n;        // overwrites h3, which is read first per element
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float ev = e[i];
        float ep;
        if (mode == 1) {
            ep = ev;
        } else if (mode == 2) {
            ep = (ev + e_prev[i]) / 2.0f;
        } else if (cnt == 1) {
            ep = (3.0f * ev - h1[i]) / 2.0f;
        } else if (cnt == 2) {
            ep = (23.0f * ev - 16.0f * h1[i] + 5.0f * h2[i]) / 12.0f;
        } else {
            ep = (55.0f * ev - 59.0f * h1[i] + 37.0f * h2[i] - 9.0f * h3[i]) / 24.0f;
        }
        const float xv = x[i];
        const float xn = xv + da * (c1 * xv - c2 * ep);
        if (mode == 1) {
            x_out[i] = xn;
        } else {
            x[i] = xn;
            hnew[i] = ev;
        }
    }
}

// advances the loop state after a step (its own launch: every block of ds_plms_kernel reads the state) and writes the
// timestep slot of the next denoiser evaluation
__global__ void ds_plms_advance_kernel(int* st, int interval, float* __restrict__ t_slot, int B) {
    __shared__ int tn;
    if (threadIdx.x == 0) {
        tn = st[0] - interval;
        st[0] = tn;
        st[1] = st[1] + 1;
        st[2] = (st[2] + 1) % 3;
    }
    __syncthreads();
    if ((int)threadIdx.x < B) t_slot[threadIdx.x] = (float)max(tn, 0);
}

__global__ void ds_fill_kernel(float* __restrict__ p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

#define DS_LAUNCH(kern, n, ...)                                                            \
    if (ctx.ws.dry) return;                                                                \
    ProfScope prof(ctx, #kern, 0.0, 8.0 * (double)(n));                                    \
    hipLaunchKernelGGL(kern, grid_for(n), dim3(256), 0, ctx.stream, __VA_ARGS__);          \
    MAA_HIP(hipGetLastError())

void launch_ds_pos_emb(const Ctx& ctx, const float* t, int B, int dim, float* out) {
    DS_LAUNCH(ds_pos_emb_kernel, (long long)B * dim / 2, t, B, dim, out);
}
void launch_ds_mish(const Ctx& ctx, const float* x, long long n, float* out) { DS_LAUNCH(ds_mish_kernel, n, x, n, out); }
void launch_ds_add_step(const Ctx& ctx, const float* x, const float* step, int ld_step, long long rows, int T, int C,
                        float* out) {
    DS_LAUNCH(ds_add_step_kernel, rows * C, x, step, ld_step, rows, T, C, out);
}
void launch_ds_gate(const Ctx& ctx, const float* y, long long rows, int C, float* z) {
    DS_LAUNCH(ds_gate_kernel, rows * C, y, rows, C, z);
}
void launch_ds_residual(const Ctx& ctx, const float* y2, long long rows, int T, int C, float* x, float* skip, int first,
                        const float* next_step, int ld_step, float* xin) {
    DS_LAUNCH(ds_residual_kernel, rows * C, y2, rows, T, C, x, skip, first, next_step, ld_step, xin);
}
void launch_ds_plms(const Ctx& ctx, float* x, const float* e, const float* e_prev, float* hist, long long n,
                    const float* ac, int interval, int* st, int mode, float* x_out) {
    DS_LAUNCH(ds_plms_kernel, n, x, e, e_prev, hist, n, ac, interval, st, mode, x_out);
}
void launch_ds_plms_advance(const Ctx& ctx, int* st, int interval, float* t_slot, int B) {
    if (ctx.ws.dry) return;
    MAA_CHECK(B <= 256, "plms: at most 256 samples per call");
    hipLaunchKernelGGL(ds_plms_advance_kernel, dim3(1), dim3(256), 0, ctx.stream, st, interval, t_slot, B);
    MAA_HIP(hipGetLastError());
}
void launch_ds_fill(const Ctx& ctx, float* p, int n, float v) {
    if (ctx.ws.dry) return;
    hipLaunchKernelGGL(ds_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx.stream, p, n, v);
    MAA_HIP(hipGetLastError());
}

}  // namespace maa
