// Internal declarations of libaudiogpt_mi355x: context, workspace arena, kernel launchers.
// gfx950 only.  Activations are channels-last fp32: images [B, H, W, C], sequences [B, L, C].
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace maa {

// ------------------------------------------------------------------------------------------ errors
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
void set_last_error(const std::string& s);

#define MAA_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            throw ::maa::Error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
    } while (0)
#define MAA_CHECK(cond, msg)                                                                       \
    do {                                                                                           \
        if (!(cond)) throw ::maa::Error(std::string("check failed: ") + #cond + " -- " + (msg));   \
    } while (0)

// ------------------------------------------------------------------------------------------ arena
// Bump allocator over one hipMalloc'd slab.  Nothing is allocated while a forward runs (hipGraph
// capture safe): the slab is grown only between forwards (`reserve`).
class Arena {
public:
    ~Arena();
    void reserve(size_t bytes);                 // (re)allocate the slab if smaller; not during capture
    void reset() { off_ = 0; run_high_ = 0; }
    size_t mark() const { return off_; }
    void release(size_t m) { off_ = m; }
    float* alloc_f(size_t n_floats);
    size_t capacity() const { return cap_; }
    const void* base() const { return base_; }
    size_t high_water() const { return high_; }
    size_t mark_high() const { return run_high_; }   // high-water since the last reset()
    // when true, alloc only counts (dry run to size the slab)
    bool dry = false;

private:
    char* base_ = nullptr;
    size_t cap_ = 0, off_ = 0, high_ = 0, run_high_ = 0;
};

// Optional per-launch timing (hipEvents on the context's stream).  Off unless maa_prof_begin was called;
// never active inside a graph capture.
struct ProfRow {
    std::string name;
    long long launches = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0;
};
class Profiler {
public:
    ~Profiler();
    struct Pending {
        std::string name;
        double flops, bytes;
        hipEvent_t e0, e1;
    };
    hipEvent_t get_event();
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    size_t next = 0;
    int detail = 0;      // 1: igemm rows are keyed by problem shape
    std::vector<ProfRow> collect(hipStream_t stream);   // synchronises, aggregates by name, clears
};
struct Ctx;
struct ProfScope {     // records an event pair around the launches issued during its lifetime
    ProfScope(const Ctx& ctx, const char* name, double flops, double bytes);
    ~ProfScope();
    const Ctx* ctx_ = nullptr;
    size_t idx_ = 0;
};

// XCD-contiguous work order (device code).  Workgroup `bid` of a 1-D grid of `n` runs on XCD bid % 8 (observed placement, used for
// speed only: MI355X_MICROARCH.md); XCD x takes the x-th contiguous eighth of the work items, workgroup by workgroup -- the order every
// implicit-GEMM engine gives its tiles (M-major), so that the rows a workgroup of a normalisation / attention / reduce launch reads
// were written, and the rows it writes will be read, by workgroups of the SAME XCD: its L2 (4 MB, not shared between XCDs) instead
// of the fabric.  `on` = 0: the identity (A/B).
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_contiguous(int bid, int n, int on) {
    if (!on) return bid;
    const int xcd = bid & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device); safe from several launching threads
void ensure_dynamic_lds(const void* kernel, int device, int bytes);
int device_cu_count(int device);      // multiProcessorCount, cached per device

// grow-only device buffer owned by a context / model (growing synchronises the stream first: never inside a capture)
struct DevSlab {
    void* p = nullptr;
    size_t cap = 0;
    ~DevSlab();
    void* get(size_t bytes, hipStream_t stream);
    DevSlab() = default;
    DevSlab(const DevSlab&) = delete;
    DevSlab& operator=(const DevSlab&) = delete;
};

// The captured DDIM step of the last sample() call on a context, kept across calls: a call whose step would launch the
// same kernels on the same addresses with the same arguments (same model, shapes, guidance, buffers) replays it instead
// of capturing and instantiating again.  `key` lists everything the step's launches depend on.
struct StepGraph {
    std::vector<unsigned long long> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void clear();
    ~StepGraph() { clear(); }
    StepGraph() = default;
    StepGraph(const StepGraph&) = delete;
    StepGraph& operator=(const StepGraph&) = delete;
};

// Test / A-B switches: every one of them is parsed in Tuning::load (runtime.cpp) -- the only place the library reads the
// environment -- when a context is created and again by maa_ctx_reload_tuning; a kept DDIM graph is dropped on reload.
struct Tuning {
    std::string dma2;                       // MAA_DMA2 = "off" | "0,4,1,S[,kmin[,kmax]]": the split-K LDS-DMA engine's policy override
    std::string pp, pp1;                    // MAA_PP / MAA_PP1 = "off" | "bn,S": tile width / K slices of the ping-pong engine (3x3 form, 1x1 form)
    int pp_s_narrow = 2, pp_s_wide = 4;     // MAA_PP_S = "a,b": K slices of the ping-pong engine's 3x3 layers with < 4 / >= 4 N tiles (10x78 / 5x39 in the UNet)
    bool up2 = true;                        // MAA_UP2=0: Upsample + conv3x3 through the virtual-upsample gather (rounds 1-5) instead of the four-phase form
    bool pp_tile_major = false;             // MAA_PP_TILE_MAJOR=1: round 5's item order (a tile's K slices are neighbours); 0: slice-major (an XCD streams 1 / S of the weights)
    bool op_presplit = false;               // MAA_OP_PRESPLIT=1: the maa_op_* test entry points hand activations over as split32
    bool no_dma = false;                    // MAA_NO_DMA: every bf16 contraction on the register-staged engine (bit-identity tests)
    int halo = 2;                           // MAA_HALO = "off": the narrow vocoder stages through the implicit GEMM; "single": their MRF pairs as two halo launches; default: fused pairs (bit-identity tests)
    bool cfg_shared = true;                 // MAA_CFG_SHARED=0: a guided DDIM step on one stream evaluates both halves of cat([x] * 2) in full (rounds 1-6a) instead of computing the layers before the first cross-attention once
    bool gn_two_pass = false;               // MAA_GN_TWO_PASS=1: GroupNorm as the statistics + apply launches everywhere (the VAE's large images always take them; tests)
    void load();
};

struct Ctx {
    Ctx() = default;
    Ctx(const Ctx&) = delete;
    Ctx& operator=(const Ctx&) = delete;
    ~Ctx();
    // Second lane of a classifier-free-guidance DDIM step (ddim.cpp): the unconditional and the conditional half of the UNet
    // batch are independent, so the step forks after the UNet input is built -- half 0 on this context's stream, half 1 on the
    // lane's own stream and workspace -- and joins before the combine + update kernel.  Captured, they are two branches of the
    // step graph.  Created on first use (side_lane), owned by this context; shares the zero page, the device and the tuning.
    Ctx* side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // How many contexts' launches the caller keeps in flight on this device (maa_ctx_set_concurrency).  >= 3 = the chip is kept
    // full from outside: a launch then costs the sum of its workgroups' time, not the rounds its own grid makes -- one stream per
    // guided DDIM step and igemm tiles by least total workgroup time; 1 or 2 = this context (nearly) owns the GPU: two CFG lanes
    // (+4 % with one context, +3.8 % with two, -24 % with three: profiles/r5/r5_call1_cfg_lanes_ab.txt, r5_call2_mixed_cfg_lanes_ab.txt),
    // tiles by least launch time; -1 (default) = not told: treated as 1 -- an agent that loads three tools and runs one at a time has
    // three idle contexts, so the number of live contexts says nothing about what is in flight.  Every choice here is bit-identical.
    int concurrency = -1;
    bool kept_full() const { return concurrency >= 3; }
    int cfg_split = -1;       // -1: two lanes unless kept_full(); 0 / 1: maa_ctx_set_cfg_split
    bool split_cfg() const { return cfg_split < 0 ? !kept_full() : cfg_split != 0; }
    Tuning tune;
    StepGraph ddim_graph;
    DevSlab sampler_scratch;  // DDIM loop state (tables, step slots, UNet input, eps): reused by every sample() call
    Profiler* prof = nullptr;
    float* zeros = nullptr;   // 256 B zero page (device), source of masked tile loads
    int device = 0;
    hipStream_t stream = nullptr;
    mutable Arena ws;    // (mutable: a launch may borrow scratch, e.g. split-K slabs, and give it back before it returns)
    int dtype = 0;   // 0 = fp32 (exact-f32 MFMA); 1 = bf16 MFMA operands, fp32 accumulate/storage
};

Ctx& side_lane(Ctx& ctx);      // the context's second lane (created on first use: stream, events)

// ------------------------------------------------------------------------------------------ igemm
// Out[m, n] = epilogue( alpha * sum_k A[m, k] * B[k, n] ),  m = (b, oy, ox) output position,
// k = (ky, kx, ci) with ci running over channels of src1 then src2 (virtual concat).
struct IGemm {
    // A operand: gathered from one or two channels-last sources
    const float* a1 = nullptr;
    const float* a2 = nullptr;
    int lda1 = 0, lda2 = 0;          // elements between consecutive spatial positions
    int C1 = 0, C2 = 0;              // channels taken from src1 / src2
    int Hin = 1, Win = 1, Hout = 1, Wout = 1;
    int KH = 1, KW = 1, sh = 1, sw = 1, ph = 0, pw = 0, dh = 1, dw = 1;
    int up = 0;                      // 1: source is read through a virtual nearest-2x upsample
    int a_split = 0;                 // A rows are split32 lines ([32 bf16 hi | 32 bf16 lo] per 32 channels, pitch lda1 floats)
    int a_act = 0;                   // 0 none, 1 leaky-relu(a_slope)   (applied while staging an fp32 A)
    float a_slope = 0.f;
    // B operand
    const float* b = nullptr;
    int ldb = 0;
    int b_nk = 0;                    // 0: B stored [K][N] (packed weights, V);  1: stored [N][K] (K^T)
    int b_split = 0;                 // B rows ([N][K]) are split32 lines, pitch ldb floats
    // dims
    int M = 0, N = 0, K = 0;
    // batching over blockIdx.z: z -> (zo, zi) = (z / zin, z % zin)
    int Z = 1, zin = 1;
    long long a_so = 0, a_si = 0, b_so = 0, b_si = 0, c_so = 0, c_si = 0;
    // epilogue
    float alpha = 1.f;
    const float* bias = nullptr;     // [N]
    const float* rowadd = nullptr;   // [B][ld_rowadd]  (time-embedding add)
    int ld_rowadd = 0;               // row index = m / (Hout*Wout)
    const float* res = nullptr;      // residual, same indexing as c
    int ldr = 0;
    int geglu = 0;                   // value/gate column interleave (see pack.cpp), writes N/2 columns
    int act = 0;                     // 0 none, 1 tanh, 2 relu, 3 gelu (erf), 4 leaky-relu(act_slope)
    float act_slope = 0.f;
    float out_scale = 1.f;           // applied after bias/residual/act
    int accumulate = 0;              // c += value instead of c = value
    int c_split = 0;                 // write the output as split32 lines (bf16 engine only; no accumulate, Z == 1)
    float* c = nullptr;
    int ldc = 0;
    // optional second output: leaky-relu(c2_slope) of the final value as split32 lines, same indexing as c with pitch ldc2 --
    // the next convolution's pre-activated, pre-split input (the vocoders' MRF chains)
    float* c2 = nullptr;
    int ldc2 = 0;
    float c2_slope = 1.f;
    int no_pair = 0;                 // (A/B, retired) plain 4-byte fp32 stores in the epilogue
    const float* zeros = nullptr;    // >= 16 B of zeros in device memory (filled in by launch_igemm)
    int m_fastest = 0;               // tile order inside an XCD's range: 1 = M-tiles fastest (MAA_TILE_ORDER=1: weights are
                                     // then fetched once chip-wide, but the conv's A re-reads lose their L2: +4 % step time)
};
void launch_igemm(const Ctx& ctx, const IGemm& p);
// Tile choice shared by the fp32 and bf16 engines: 0 = 128x128, 1 = 128x64, 2 = 64x64 (3 = 256x32 is chosen by
// the callers for N <= 32).  Cost = CU-rounds x tile area / (tile efficiency x latency hiding at that many
// co-resident blocks per CU); knobs can be overridden for tuning with MAA_TILE_EFF / MAA_CONC_EFF / MAA_FORCE_CFG.
int choose_tile(long long M, long long N, int Z, bool bf16, int mode = 0);      // mode 1: least total workgroup time (several contexts keep the chip full)
bool launch_igemm_bf16(const Ctx& ctx, const IGemm& p, int terms);   // false: not eligible, use the fp32 kernel
// bf16x3 with LDS-DMA tile copies, both operands split32 (igemm_dma.hip); called by launch_igemm_bf16
int igemm_dma_tile(const IGemm& p, int cfg);      // tile the DMA engine runs for the generic choice `cfg`
void launch_igemm_dma(const Ctx& ctx, const IGemm& p, int cfg, int Nb);
// second LDS-DMA engine (igemm_dma2.hip): 128x128 / 256x128 tiles, 64x64 outputs per wave, split-K finished by a
// fixed-order reduce kernel.  `takes` and the slab size depend on the layer (K, packed N) only, never on M.
struct Dma2Plan {
    int cfg = -1;       // -1: not taken; 0: the 128x128 tile (4 waves of 64x64)
    int ns = 4, pipe = 1, S = 1;      // LDS stages, in-wave pipelining, K slices
};
Dma2Plan igemm_dma2_plan(const Ctx& ctx, const IGemm& p);
size_t igemm_dma2_workspace_floats(const IGemm& p, const Dma2Plan& pl);      // 0: no split-K for this problem
const char* igemm_dma2_name(const Dma2Plan& pl, int terms);
void launch_igemm_dma2(const Ctx& ctx, const IGemm& p, int Nb, const Dma2Plan& pl, float* part);

// adds the S slabs a split-K engine wrote in fragment order ([slice][tile][MI NI blocks][4 quads][NTH threads][4 floats]) in
// slice order and applies the epilogue (igemm_dma2.hip); WGN / MI / NI / NTH describe the GEMM workgroup's wave geometry
void launch_splitk_reduce(const Ctx& ctx, const IGemm& p, const float* part, int S, int tiles, int ntiles, int Nb, int BM,
                          int BN, int WGN, int MI, int NI, int NTH);
// third LDS-DMA engine (igemm_pp.hip): 3x3 convolutions with the A operand halo-staged once per channel chunk and two wave
// groups alternating matrix / memory phases.  bn = 0: not taken.  Tile width and K slices depend on the layer only.
struct PPPlan {
    int bn = 0, S = 1;
};
PPPlan igemm_pp_plan(const Ctx& ctx, const IGemm& p);
size_t igemm_pp_workspace_floats(const IGemm& p, const PPPlan& pl);
const char* igemm_pp_name(const PPPlan& pl, int terms);
void launch_igemm_pp(const Ctx& ctx, const IGemm& p, int Nb, const PPPlan& pl, float* part);
// ... and its 1x1 / Linear form (no halo; A and the weights share one ring)
PPPlan igemm_pp1_plan(const Ctx& ctx, const IGemm& p);
size_t igemm_pp1_workspace_floats(const IGemm& p, const PPPlan& pl);
const char* igemm_pp1_name(const PPPlan& pl, int terms);
void launch_igemm_pp1(const Ctx& ctx, const IGemm& p, int Nb, const PPPlan& pl, float* part);

// ------------------------------------------------------------------------------------------ norms etc.
// GroupNorm(32 groups) over a channels-last tensor given as a virtual concat of two sources; writes
// a dense [B, HW, C1+C2] tensor.  silu: fuse x*sigmoid(x).
// out_split = 1: the output rows are written as split32 lines (see IGemm::a_split) for a bf16-engine
// consumer instead of fp32 (same byte size).  Takes 2*B*C floats of scratch (per-sample channel scale/shift) from the arena.
void launch_groupnorm(Ctx& ctx, const float* x1, int ld1, int C1, const float* x2, int ld2, int C2, int B, int HW,
                      int groups, const float* gamma, const float* beta, float eps, int silu, float* out,
                      int out_split = 0, float* raw_split = nullptr);
// raw_split: optional second output [B, HW, C1+C2] -- the UN-normalised (x1 | x2) rows as split32 lines (C % 32 == 0)
void launch_layernorm(const Ctx& ctx, const float* x, long long rows, int C, const float* gamma, const float* beta,
                      float eps, float* out, int out_split = 0);
// fp32 [rows, C] -> split32 rows of the same pitch (C % 32 == 0): tests and micro-benchmarks of the engines that take
// pre-split activations (in the models the normalisations write this form directly)
// "nearest-2x upsample + 3x3 convolution" as four 2x2 phase convolutions in one launch of the ping-pong engine (igemm_pp.hip);
// planes [4][B H W][C] -> image [B, 2H, 2W, C] (misc.hip)
bool launch_igemm_pp_up2(const Ctx& ctx, const IGemm& p, int Nb, long long b_phase);
// conv3x3 (stride 1, pad 1) to N <= 4 channels from split32 rows, output in NCHW (misc.hip; bf16x3 mode); w4 = [9][C] float4 of
// weights in [tap][channel % 8][channel / 8] order (WeightStore::pack_narrow3x3), bias [4].  false: not applicable -> the implicit GEMM
bool launch_narrow_conv3x3(const Ctx& ctx, const float* a_split, int lda, int B, int H, int W, int C, const float* w4, const float* bias,
                           int N, float* out_nchw);
void launch_pixel_shuffle2(const Ctx& ctx, const float* planes, int B, int H, int W, int C, float* out);
void launch_split32_pack(const Ctx& ctx, const float* x, long long rows, int C, float* out, float slope = 1.f);
void launch_split32_unpack(const Ctx& ctx, const float* x, long long rows, int C, float* out);      // hi + lo back to fp32 (tests)
// fused softmax(alpha q k^T) v for the bf16 precision modes; false = shape not covered, use the GEMM path
bool launch_flash_attention(const Ctx& ctx, const float* q, int ldq, int hsq, const float* k, int ldk, int hsk,
                            const float* v, int ldv, int hsv, int B, int heads, int dh, int Nq, int Nk, float alpha,
                            float* out, int ldo, int out_split = 0, int causal = 0);
bool flash_attention_covers(const Ctx& ctx, int dh);    // head widths launch_flash_attention takes in this mode
// in-place row softmax over `cols` columns of a [rows, ld] matrix; columns [cols, ld) are zeroed
void launch_softmax(const Ctx& ctx, float* s, long long rows, int cols, int ld, int causal_nq = 0);

// ------------------------------------------------------------------------------------------ elementwise
void launch_timestep_embedding(const Ctx& ctx, const float* t, int B, int dim, float* out);   // [cos | sin]
void launch_silu(const Ctx& ctx, const float* x, long long n, float* out);
void launch_add(const Ctx& ctx, const float* a, const float* b, long long n, float* out);
void launch_nchw_to_nhwc(const Ctx& ctx, const float* x, int B, int C, int HW, float* out);
void launch_nhwc_to_nchw(const Ctx& ctx, const float* x, int B, int C, int HW, float* out, int ld_in);
void launch_avgpool2(const Ctx& ctx, const float* x, int B, int H, int W, int C, float* out);
void launch_upsample2(const Ctx& ctx, const float* x, int B, int H, int W, int C, float* out);
// out[0 .. n) = out[n .. 2n) = x[0 .. n)   (n % 4 == 0): the two halves of a guided step's batch leave their shared prefix
void launch_dup_half(const Ctx& ctx, const float* x, long long n, float* out);
void launch_scale(const Ctx& ctx, const float* x, long long n, float s, float* out);
// eps = eu + scale*(ec - eu); x0 = (x - somat*eps)/sqrt(a_t); x' = sqrt(a_prev)*x0 + sqrt(1-a_prev-sig^2)*eps
// coef = device pointer to {a_t, a_prev, sigma, sqrt_one_minus_at}; eps_c may be null (no CFG)
// step (optional): device DDIM index, decremented for the next step
void launch_ddim_update(const Ctx& ctx, const float* x, const float* eps_u, const float* eps_c, float scale,
                        const float* coef, long long n, float* x_prev, float* pred_x0, int* step = nullptr);
// UNet input of a step (latents duplicated for CFG when nB = 2B, or concatenated with the inpaint conditioning) and the
// step's timestep / coefficient slots (8 floats per DDIM index: a_t, a_prev, sigma, sqrt(1-a_t), sqrt(ac_t), sqrt(1-ac_t), log
// slot, 0), selected from device tables by the device index *step; mask / x0 / noise_q: the mask blend of ddim.py:147-150
void launch_ddim_prepare(const Ctx& ctx, const float* x, const float* concat, int B, int nB, long long per,
                         long long per_c, const float* tab_t, const float* tab_coef, const int* step, float* xin,
                         float* cur_t, float* cur_coef, const float* mask = nullptr, const float* x0 = nullptr,
                         const float* noise_q = nullptr, int S = 0, const float* emb_tab = nullptr, int emb_w = 0,
                         float* cur_emb = nullptr);
// emb_tab [S][emb_w] / cur_emb [emb_w]: the step's precomputed ResBlock time-embedding row is copied into its fixed slot
// the loop's update: x from the step's UNet input, optional sigma_t * noise * temperature, logged intermediates, index - 1
void launch_ddim_step(const Ctx& ctx, const float* xin, long long per, long long per_in, const float* eps_u, const float* eps_c,
                      float scale, const float* coef, long long n, float* x_prev, const float* noise_p, float temperature, int S,
                      float* log_x, float* log_x0, int* step);
// BigVGAN Activation1d on [B, L, C]: up2 FIR -> snake -> down2 FIR (replicate padding)
void launch_snake_aa(const Ctx& ctx, const float* x, int B, int L, int C, const float* inv_beta, const float* alpha,
                     float* out);
// NSF harmonic source (nsf.hip): f0 [B, T] -> sines [B, T*hop, harmonics+1] (scratch) -> har [B, T*hop]
void launch_nsf_source(const Ctx& ctx, const float* f0, int B, int T, int hop, float sampling_rate, const float* rand_ini,
                       const float* noise, int harmonics, const float* w, const float* bias, float* sines, float* har);
// out[i, j] = scale * <audio[i], text[j]> (spectral.hip)
void launch_similarity(const Ctx& ctx, const float* audio, const float* text, int Na, int Nt, int D, float scale, float* out);
void launch_leaky(const Ctx& ctx, const float* x, long long n, float slope, float* out);
void launch_clamp_affine(const Ctx& ctx, const float* x, long long n, float mul, float add, float lo, float hi,
                         float* out);
void launch_spec_from_mel(const Ctx& ctx, const float* mel, long long n, float* spec);      // clamp((mel + 1) / 2, 0, 1)

// box calibration (calib.hip): kind 0 -> dense bf16 MFMA TFLOP/s of a fixed register-only loop, kind 1 -> GB/s of a 256 MiB copy
double calib_run(const Ctx& ctx, int kind);

// ------------------------------------------------------------------------------------------ weights
struct HostTensor {
    const float* data = nullptr;
    std::vector<long long> shape;
    long long numel() const {
        long long n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};
using StateDict = std::map<std::string, HostTensor>;

// device-resident packed weight for the igemm B operand, fp32:
//   nk = 0: [K][Npad] (row pitch ld = Npad)    -- exact-fp32 mode
//   nk = 1: [Npad][Kpad] (row pitch ld = K rounded up to 4, zero padded) -- bf16 modes (k-contiguous operands)
struct PackedW {
    int phase_rows = 0;     // pack_conv_up2: packed rows between the four phases' weights (0: an ordinary weight)
    float* w = nullptr;
    float* bias = nullptr;  // [Npad] or null
    int K = 0, N = 0, Npad = 0;
    int ld = 0, nk = 0;
    int split = 0;          // 1: rows of w are split32 lines (row pitch ld floats)
};

// Conv1d(C, C, k, dilation) with "same" padding for the narrow vocoder stages (C = 32 / 64), bf16x3: the input tile is
// staged once in LDS and every tap reads it at a row offset (halo_conv1d.hip).  false: not covered, use the implicit GEMM.
bool launch_halo_conv1d(const Ctx& ctx, const float* x, int B, int L, int C, const PackedW& w, int k, int dil, float slope,
                        const float* res, float out_scale, int accumulate, float* out);
// the pair c2(leaky(c1(leaky(x)))) + res of an MRF resblock in one launch, the intermediate tensor kept in LDS (halo_conv1d.hip)
bool launch_halo_pair(const Ctx& ctx, const float* x, int B, int L, int C, const PackedW& w1, int k1, int d1, float slope1,
                      const PackedW& w2, int k2, int d2, float slope2, const float* res, float out_scale, int accumulate,
                      float* out);

class WeightStore {
public:
    explicit WeightStore(bool nk_layout = false) : nk_(nk_layout) {}
    ~WeightStore();
    // upload a host [K][Npad] matrix in this store's layout and fill w / ld / nk
    // bf16_ok: every use of this weight is eligible for the bf16 engine (then it is stored pre-split)
    void finish(PackedW& pw, const std::vector<float>& kn, bool bf16_ok = false);
    // the four 2x2 phase weights of "nearest-2x upsample + conv3x3" (taps reading the same source pixel summed); w == nullptr when
    // this store's layout / the channel count does not allow the phase form
    PackedW pack_conv_up2(const StateDict& sd, const std::string& wname, const std::string& bname);
    // conv3x3 with Cout <= 4 for launch_narrow_conv3x3: w = [9][Cin] float4 (tap, channel; the outputs in x y z w), bias = [4] floats
    PackedW pack_narrow3x3(const StateDict& sd, const std::string& wname, const std::string& bname);
    void* upload_raw(const void* host, size_t bytes);
    float* upload(const std::vector<float>& host);
    // conv / linear weight [Cout][Cin][KH][KW] (linear: KH=KW=1) -> [ (ky,kx,ci) ][Cout pad 32]
    PackedW pack_conv(const StateDict& sd, const std::string& wname, const std::string& bname, int KH, int KW);
    // several linears sharing the input, concatenated on the output axis
    PackedW pack_concat(const StateDict& sd, const std::vector<std::string>& wnames,
                        const std::vector<std::string>& bnames);
    // GEGLU projection: value/gate columns interleaved in groups of 32 (see igemm epilogue)
    PackedW pack_geglu(const StateDict& sd, const std::string& wname, const std::string& bname);
    // ConvTranspose1d weight [Cin][Cout][k], stride s, padding p: polyphase group `carry` (0/1)
    PackedW pack_convtr_phase(const StateDict& sd, const std::string& wname, const std::string& bname, int stride,
                              int pad, int carry, int* r_start, int* r_count);
    float* vec(const StateDict& sd, const std::string& name);
    size_t bytes() const { return bytes_; }

private:
    std::vector<void*> bufs_;
    size_t bytes_ = 0;
    bool nk_ = false;
};

const HostTensor& get(const StateDict& sd, const std::string& name);
bool has(const StateDict& sd, const std::string& name);

}  // namespace maa
