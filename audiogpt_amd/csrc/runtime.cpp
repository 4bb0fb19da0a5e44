// Runtime support: workspace arena, weight store and host-side weight repacking.
//
// Packed layout for every igemm B operand: row index k = (ky*KW + kx)*Cin + ci, column = output channel,
// zero-padded to a multiple of 32 columns.  Reference weight layouts being repacked:
//   Conv2d [Cout][Cin][KH][KW], Conv1d [Cout][Cin][k], Linear [Cout][Cin]   (torch.nn defaults)
//   ConvTranspose1d [Cin][Cout][k]  (NeuralSeq/modules/hifigan/hifigan.py:121-125)
#include "maa_internal.h"

#include <mutex>

#include <cstdlib>
#include <cstring>

extern char** environ;

namespace maa {

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
const char* last_error_cstr() { return g_last_error.c_str(); }

// ------------------------------------------------------------------------------------------ Arena
Arena::~Arena() {
    if (base_) (void)hipFree(base_);
}
void Arena::reserve(size_t bytes) {
    if (bytes <= cap_) return;
    if (base_) MAA_HIP(hipFree(base_));
    base_ = nullptr;
    cap_ = 0;
    MAA_HIP(hipMalloc(reinterpret_cast<void**>(&base_), bytes));
    cap_ = bytes;
}
float* Arena::alloc_f(size_t n_floats) {
    const size_t bytes = (n_floats * sizeof(float) + 255) / 256 * 256;
    const size_t at = off_;
    off_ += bytes;
    if (off_ > high_) high_ = off_;
    if (off_ > run_high_) run_high_ = off_;
    if (dry) return reinterpret_cast<float*>(static_cast<uintptr_t>(0x1000) + at);   // never dereferenced
    if (off_ > cap_) throw Error("workspace arena exhausted (" + std::to_string(off_) + " > " + std::to_string(cap_) + ")");
    return reinterpret_cast<float*>(base_ + at);
}

DevSlab::~DevSlab() {
    if (p) (void)hipFree(p);
}
void* DevSlab::get(size_t bytes, hipStream_t stream) {
    if (bytes > cap) {
        if (p) {
            MAA_HIP(hipStreamSynchronize(stream));      // earlier launches may still read the old buffer
            MAA_HIP(hipFree(p));
            p = nullptr;
            cap = 0;
        }
        const size_t want = bytes + bytes / 4;
        MAA_HIP(hipMalloc(&p, want));
        cap = want;
    }
    return p;
}

// ------------------------------------------------------------------------------------------ Profiler
Profiler::~Profiler() {
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
}
hipEvent_t Profiler::get_event() {
    if (next == pool.size()) {
        hipEvent_t e;
        MAA_HIP(hipEventCreate(&e));
        pool.push_back(e);
    }
    return pool[next++];
}
std::vector<ProfRow> Profiler::collect(hipStream_t stream) {
    MAA_HIP(hipStreamSynchronize(stream));
    std::map<std::string, ProfRow> agg;
    for (auto& p : pending) {
        float ms = 0.f;
        MAA_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
        ProfRow& r = agg[p.name];
        r.name = p.name;
        r.launches += 1;
        r.ms += ms;
        r.flops += p.flops;
        r.bytes += p.bytes;
    }
    pending.clear();
    next = 0;
    std::vector<ProfRow> out;
    for (auto& kv : agg) out.push_back(kv.second);
    return out;
}
ProfScope::ProfScope(const Ctx& ctx, const char* name, double flops, double bytes) {
    if (!ctx.prof || ctx.ws.dry) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx.stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
    ctx_ = &ctx;
    Profiler::Pending p;
    p.name = name;
    p.flops = flops;
    p.bytes = bytes;
    p.e0 = ctx.prof->get_event();
    p.e1 = ctx.prof->get_event();
    (void)hipEventRecord(p.e0, ctx.stream);
    idx_ = ctx.prof->pending.size();
    ctx.prof->pending.push_back(p);
}
ProfScope::~ProfScope() {
    if (ctx_) (void)hipEventRecord(ctx_->prof->pending[idx_].e1, ctx_->stream);
}

// ------------------------------------------------------------------------------------------ tile choice
namespace {
struct TileKnobs {
    double eff[3] = {1.00, 0.92, 0.80};       // per-block efficiency of 128x128 / 128x64 / 64x64
    double conc[3] = {0.55, 0.85, 1.00};      // latency hiding with 1 / 2 / >=3 co-resident blocks per CU
};
}  // namespace
int choose_tile(long long M, long long N, int Z, bool bf16, int mode) {
    static const TileKnobs k;
    static const int bm[3] = {128, 128, 64}, bn[3] = {128, 64, 64};
    const int max_occ[3] = {bf16 ? 2 : 3, bf16 ? 2 : 4, 4};   // blocks per CU allowed by LDS / registers
    int best = 0;
    double best_cost = 1e300;
    for (int c = 0; c < 3; ++c) {
        const long long blocks = ((M + bm[c] - 1) / bm[c]) * ((N + bn[c] - 1) / bn[c]) * Z;
        const long long per_cu = (blocks + 255) / 256;
        const int co = (int)(per_cu < max_occ[c] ? per_cu : max_occ[c]);
        // mode 1: the chip is kept full from outside (other contexts' launches run on the CUs this one leaves idle), so what a
        // launch costs is the sum of its workgroups' time, padding included -- not the rounds its own grid makes
        const double cost = mode == 1 ? (double)blocks * bm[c] * bn[c] / k.eff[c]
                                      : (double)per_cu * bm[c] * bn[c] / (k.eff[c] * k.conc[co >= 3 ? 2 : co - 1]);
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------ StateDict helpers
const HostTensor& get(const StateDict& sd, const std::string& name) {
    auto it = sd.find(name);
    if (it == sd.end()) throw Error("missing weight tensor: " + name);
    return it->second;
}
bool has(const StateDict& sd, const std::string& name) { return sd.find(name) != sd.end(); }

// The ONE place the library reads its environment: when a context is created and in maa_ctx_reload_tuning.  Every knob is a
// test / A-B switch with the product's behaviour as its default; -DMAA_NO_TUNING compiles the parsing out (a deployment build
// whose behaviour cannot be changed from outside).
void Tuning::load() {
#ifdef MAA_NO_TUNING
    return;
#endif
    auto get_s = [](const char* name) {
        const char* e = std::getenv(name);
        return std::string(e ? e : "");
    };
    dma2 = get_s("MAA_DMA2");
    pp = get_s("MAA_PP");
    pp1 = get_s("MAA_PP1");
    pp_s_narrow = 2;
    pp_s_wide = 4;
    const std::string pss = get_s("MAA_PP_S");
    if (!pss.empty()) std::sscanf(pss.c_str(), "%d,%d", &pp_s_narrow, &pp_s_wide);
    if (pp_s_narrow < 1) pp_s_narrow = 1;
    if (pp_s_wide < 1) pp_s_wide = 1;
    pp_tile_major = get_s("MAA_PP_TILE_MAJOR") == "1";
    up2 = get_s("MAA_UP2") != "0";
    cfg_shared = get_s("MAA_CFG_SHARED") != "0";
    const std::string ps = get_s("MAA_OP_PRESPLIT");
    op_presplit = !ps.empty() && ps[0] == '1';
    no_dma = !get_s("MAA_NO_DMA").empty();
    const std::string hs = get_s("MAA_HALO");
    halo = hs == "off" ? 0 : hs == "single" ? 1 : 2;
    gn_two_pass = get_s("MAA_GN_TWO_PASS") == "1";
    // a stale override in an older round's format ("2,2,0,1": tile, stages ...) is refused here, when the context is created
    // (last, so that every other knob is in place), not by a check in the middle of a forward pass
    if (!dma2.empty() && dma2 != "off") {
        int cfg = 0, ns = 4, pipe = 1, S = 1, kmin = 0, kmax = 0;
        const int k = std::sscanf(dma2.c_str(), "%d,%d,%d,%d,%d,%d", &cfg, &ns, &pipe, &S, &kmin, &kmax);
        if (k < 4 || cfg != 0 || ns != 4 || pipe != 1 || S < 1) {
            const std::string bad = dma2;
            dma2.clear();      // the context keeps running on the default policy if the caller catches the error
            throw Error("MAA_DMA2=\"" + bad + "\": expected \"off\" or \"0,4,1,S[,kmin[,kmax]]\" (the split-K engine keeps one "
                        "instantiation: tile 0, 4 stages, pipelined; S = K slices)");
        }
    }
}

Ctx::~Ctx() {
    if (side) {
        (void)hipStreamSynchronize(side->stream);
        (void)hipStreamDestroy(side->stream);
        side->stream = nullptr;
        delete side->prof;
        side->prof = nullptr;
        delete side;
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
}
Ctx& side_lane(Ctx& ctx) {
    if (!ctx.side) {
        hipStream_t s = nullptr;
        MAA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        Ctx* sd = new Ctx;
        sd->stream = s;
        sd->device = ctx.device;
        sd->zeros = ctx.zeros;
        ctx.side = sd;
        MAA_HIP(hipEventCreateWithFlags(&ctx.ev_fork, hipEventDisableTiming));
        MAA_HIP(hipEventCreateWithFlags(&ctx.ev_join, hipEventDisableTiming));
    }
    ctx.side->tune = ctx.tune;
    ctx.side->dtype = ctx.dtype;
    ctx.side->concurrency = ctx.kept_full() ? 3 : 1;      // the lane follows its context's arrangement (it must not guess on its own)
    return *ctx.side;
}

void StepGraph::clear() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    exec = nullptr;
    graph = nullptr;
    key.clear();
}

// ------------------------------------------------------------------------------------------ per-device launch state
namespace {
std::mutex g_launch_mu;
std::map<std::pair<const void*, int>, int> g_lds_attr;      // (kernel, device) -> bytes already granted
std::map<int, int> g_cus;
}  // namespace

void ensure_dynamic_lds(const void* kernel, int device, int bytes) {
    std::lock_guard<std::mutex> lk(g_launch_mu);
    int& have = g_lds_attr[{kernel, device}];
    if (have >= bytes) return;
    MAA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
}

int device_cu_count(int device) {
    std::lock_guard<std::mutex> lk(g_launch_mu);
    int& c = g_cus[device];
    if (!c) {
        hipDeviceProp_t prop;
        MAA_HIP(hipGetDeviceProperties(&prop, device));
        c = prop.multiProcessorCount;
    }
    return c;
}

// ------------------------------------------------------------------------------------------ WeightStore
WeightStore::~WeightStore() {
    for (void* p : bufs_) (void)hipFree(p);
}
float* WeightStore::upload(const std::vector<float>& host) {
    void* d = nullptr;
    const size_t bytes = host.size() * sizeof(float);
    MAA_HIP(hipMalloc(&d, bytes ? bytes : 4));
    if (bytes) MAA_HIP(hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice));
    bufs_.push_back(d);
    bytes_ += bytes;
    return static_cast<float*>(d);
}
float* WeightStore::vec(const StateDict& sd, const std::string& name) {
    const HostTensor& t = get(sd, name);
    return upload(std::vector<float>(t.data, t.data + t.numel()));
}

static int pad32(int n) { return (n + 31) / 32 * 32; }

static inline unsigned short f2bf(float f) {      // round to nearest even, as v_cvt_pk_bf16_f32
    unsigned u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf2f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
void* WeightStore::upload_raw(const void* host, size_t bytes) {
    void* d = nullptr;
    MAA_HIP(hipMalloc(&d, bytes ? bytes : 4));
    if (bytes) MAA_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    bufs_.push_back(d);
    bytes_ += bytes;
    return d;
}

void WeightStore::finish(PackedW& pw, const std::vector<float>& kn, bool bf16_ok) {
    if (nk_ && bf16_ok) {
        // pre-split "split32" rows [Npad][K]: every 32 k of a row are one 128-byte line [32 hi | 32 lo] with
        // hi = bf16(w), lo = bf16(w - hi); the kernel copies the lines straight to LDS.  ld is in fp32 units.
        MAA_CHECK(pw.K % 32 == 0, "split32 weights need K % 32 == 0");
        const int ldk = pw.K;
        std::vector<unsigned short> t((size_t)pw.Npad * ldk * 2, 0);
        for (int k = 0; k < pw.K; ++k)
            for (int n = 0; n < pw.Npad; ++n) {
                const float w = kn[(size_t)k * pw.Npad + n];
                const unsigned short hi = f2bf(w);
                const size_t at = (size_t)n * ldk * 2 + (size_t)(k >> 5) * 64 + (k & 31);
                t[at] = hi;
                t[at + 32] = f2bf(w - bf2f(hi));
            }
        pw.w = static_cast<float*>(upload_raw(t.data(), t.size() * sizeof(unsigned short)));
        pw.ld = ldk;
        pw.nk = 1;
        pw.split = 1;
        return;
    }
    if (!nk_) {
        pw.w = upload(kn);
        pw.ld = pw.Npad;
        pw.nk = 0;
        return;
    }
    const int Kp = (pw.K + 3) / 4 * 4;
    std::vector<float> t((size_t)pw.Npad * Kp, 0.f);
    for (int k = 0; k < pw.K; ++k)
        for (int n = 0; n < pw.Npad; ++n) t[(size_t)n * Kp + k] = kn[(size_t)k * pw.Npad + n];
    pw.w = upload(t);
    pw.ld = Kp;
    pw.nk = 1;
}

PackedW WeightStore::pack_conv(const StateDict& sd, const std::string& wname, const std::string& bname, int KH,
                               int KW) {
    const HostTensor& w = get(sd, wname);
    MAA_CHECK(w.shape.size() >= 2, wname);
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1];
    MAA_CHECK(w.numel() == (long long)Cout * Cin * KH * KW, "conv weight shape " + wname);
    PackedW pw;
    pw.K = KH * KW * Cin;
    pw.N = Cout;
    pw.Npad = pad32(Cout);
    std::vector<float> h((size_t)pw.K * pw.Npad, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < KH * KW; ++t)
                h[((size_t)t * Cin + ci) * pw.Npad + co] = w.data[((size_t)co * Cin + ci) * KH * KW + t];
    finish(pw, h, Cin % 32 == 0);
    if (!bname.empty()) {
        const HostTensor& b = get(sd, bname);
        std::vector<float> hb(pw.Npad, 0.f);
        std::memcpy(hb.data(), b.data, sizeof(float) * Cout);
        pw.bias = upload(hb);
    }
    return pw;
}

PackedW WeightStore::pack_conv_up2(const StateDict& sd, const std::string& wname, const std::string& bname) {
    // out(2y + py, 2x + px) = sum_{ky, kx} w[ky][kx] . up(2y + py + ky - 1, 2x + px + kx - 1),  up(i, j) = x(i >> 1, j >> 1):
    // py = 0: ky = 0 reads row y - 1, ky = 1, 2 read row y;   py = 1: ky = 0, 1 read row y, ky = 2 reads row y + 1 (same in x).
    // Phase (py, px) is a 2x2 convolution over rows {y - 1 + py, y + py} x cols {x - 1 + px, x + px} = padding (1 - py, 1 - px),
    // tap (ty, tx) = the sum of the 3x3 taps that land on it.  K = (ty, tx, ci); the phases are stacked on the N axis.
    PackedW pw;
    const HostTensor& w = get(sd, wname);
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1];
    if (!nk_ || Cin % 32 != 0 || w.numel() != (long long)Cout * Cin * 9) return pw;
    const int np = pad32(Cout);
    pw.K = 4 * Cin;
    pw.Npad = 4 * np;
    std::vector<float> h((size_t)pw.K * pw.Npad, 0.f);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int phase = py * 2 + px;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int ty = py == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
                    const int tx = px == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
                    const int t = ty * 2 + tx;
                    for (int co = 0; co < Cout; ++co)
                        for (int ci = 0; ci < Cin; ++ci)
                            h[((size_t)t * Cin + ci) * pw.Npad + (size_t)phase * np + co] += w.data[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
                }
        }
    finish(pw, h, true);
    pw.N = Cout;
    pw.Npad = np;
    pw.phase_rows = np;
    if (!bname.empty()) {
        const HostTensor& b = get(sd, bname);
        std::vector<float> hb(np, 0.f);
        std::memcpy(hb.data(), b.data, sizeof(float) * Cout);
        pw.bias = upload(hb);
    }
    return pw;
}

PackedW WeightStore::pack_narrow3x3(const StateDict& sd, const std::string& wname, const std::string& bname) {
    PackedW pw;
    const HostTensor& w = get(sd, wname);
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1];
    if (Cout > 4 || Cin % 8 != 0 || w.numel() != (long long)Cout * Cin * 9 || bname.empty()) return pw;
    std::vector<float> h((size_t)9 * Cin * 4, 0.f), hb(4, 0.f);
    for (int t = 0; t < 9; ++t)
        for (int ci = 0; ci < Cin; ++ci)
            for (int co = 0; co < Cout; ++co)      // [tap][k = ci % 8][g = ci / 8][co]: a wave's lanes (g) read consecutive float4
                h[(((size_t)t * 8 + (ci & 7)) * (Cin / 8) + (ci >> 3)) * 4 + co] = w.data[((size_t)co * Cin + ci) * 9 + t];
    const HostTensor& b = get(sd, bname);
    std::memcpy(hb.data(), b.data, sizeof(float) * Cout);
    pw.w = upload(h);
    pw.bias = upload(hb);
    pw.K = 9 * Cin;
    pw.N = Cout;
    pw.Npad = 4;
    return pw;
}

PackedW WeightStore::pack_concat(const StateDict& sd, const std::vector<std::string>& wnames,
                                 const std::vector<std::string>& bnames) {
    int N = 0, Cin = -1;
    for (auto& n : wnames) {
        const HostTensor& w = get(sd, n);
        const int cin = (int)(w.numel() / w.shape[0]);
        MAA_CHECK(Cin < 0 || Cin == cin, "pack_concat: input width mismatch " + n);
        Cin = cin;
        N += (int)w.shape[0];
    }
    PackedW pw;
    pw.K = Cin;
    pw.N = N;
    pw.Npad = pad32(N);
    std::vector<float> h((size_t)Cin * pw.Npad, 0.f), hb(pw.Npad, 0.f);
    int col = 0;
    for (size_t i = 0; i < wnames.size(); ++i) {
        const HostTensor& w = get(sd, wnames[i]);
        const int co_n = (int)w.shape[0];
        for (int co = 0; co < co_n; ++co)
            for (int ci = 0; ci < Cin; ++ci) h[(size_t)ci * pw.Npad + col + co] = w.data[(size_t)co * Cin + ci];
        if (!bnames.empty() && !bnames[i].empty()) {
            const HostTensor& b = get(sd, bnames[i]);
            std::memcpy(hb.data() + col, b.data, sizeof(float) * co_n);
        }
        col += co_n;
    }
    finish(pw, h, Cin % 32 == 0);
    bool any_bias = false;
    for (auto& b : bnames) any_bias = any_bias || !b.empty();
    if (any_bias) pw.bias = upload(hb);
    return pw;
}

PackedW WeightStore::pack_geglu(const StateDict& sd, const std::string& wname, const std::string& bname) {
    // proj weight [2*inner][Cin]: rows [0, inner) = value, [inner, 2*inner) = gate  (attention.py:42-44)
    // packed column (g*64 + j)      <- value column g*32 + j
    //               (g*64 + 32 + j) <- gate  column g*32 + j
    const HostTensor& w = get(sd, wname);
    const HostTensor& b = get(sd, bname);
    const int N2 = (int)w.shape[0], Cin = (int)w.shape[1], inner = N2 / 2;
    MAA_CHECK(inner % 32 == 0, "geglu inner width must be a multiple of 32");
    PackedW pw;
    pw.K = Cin;
    pw.N = inner;       // output columns
    pw.Npad = N2;       // packed columns
    std::vector<float> h((size_t)Cin * N2), hb(N2);
    for (int j = 0; j < inner; ++j) {
        const int g = j / 32, jj = j % 32;
        const int cv = g * 64 + jj, cg = g * 64 + 32 + jj;
        for (int ci = 0; ci < Cin; ++ci) {
            h[(size_t)ci * N2 + cv] = w.data[(size_t)j * Cin + ci];
            h[(size_t)ci * N2 + cg] = w.data[(size_t)(inner + j) * Cin + ci];
        }
        hb[cv] = b.data[j];
        hb[cg] = b.data[inner + j];
    }
    finish(pw, h, Cin % 32 == 0);
    pw.bias = upload(hb);
    return pw;
}

PackedW WeightStore::pack_convtr_phase(const StateDict& sd, const std::string& wname, const std::string& bname,
                                       int stride, int pad, int carry, int* r_start, int* r_count) {
    // out[s*j + r] = sum_u sum_ci x[j + c - u][ci] * w[ci][co][phi + s*u],  phi = (r+pad) % s, c = (r+pad) / s,
    // u in [0, k/s).  All r with the same carry c share the input taps {j+c-u}; they are stacked on the
    // output-column axis: column = (r - r_start)*Cout + co.  Tap order in K: kx = (U-1) - u  <->  x[j + c - U + 1 + kx].
    const HostTensor& w = get(sd, wname);
    const int Cin = (int)w.shape[0], Cout = (int)w.shape[1], k = (int)w.shape[2];
    MAA_CHECK(k % stride == 0, "conv-transpose kernel must be a multiple of the stride");
    const int U = k / stride;
    std::vector<int> rs;
    for (int r = 0; r < stride; ++r)
        if ((r + pad) / stride == carry) rs.push_back(r);
    MAA_CHECK(!rs.empty(), "empty polyphase group");
    *r_start = rs.front();
    *r_count = (int)rs.size();
    MAA_CHECK(rs.back() - rs.front() + 1 == (int)rs.size(), "polyphase group not contiguous");
    PackedW pw;
    pw.K = U * Cin;
    pw.N = (int)rs.size() * Cout;
    pw.Npad = pad32(pw.N);
    std::vector<float> h((size_t)pw.K * pw.Npad, 0.f), hb(pw.Npad, 0.f);
    const HostTensor& b = get(sd, bname);
    for (size_t ri = 0; ri < rs.size(); ++ri) {
        const int phi = (rs[ri] + pad) % stride;
        for (int kx = 0; kx < U; ++kx) {
            const int u = U - 1 - kx;
            for (int ci = 0; ci < Cin; ++ci)
                for (int co = 0; co < Cout; ++co)
                    h[((size_t)kx * Cin + ci) * pw.Npad + ri * Cout + co] =
                        w.data[((size_t)ci * Cout + co) * k + phi + stride * u];
        }
        for (int co = 0; co < Cout; ++co) hb[ri * Cout + co] = b.data[co];
    }
    finish(pw, h, Cin % 32 == 0);
    pw.bias = upload(hb);
    return pw;
}

}  // namespace maa
