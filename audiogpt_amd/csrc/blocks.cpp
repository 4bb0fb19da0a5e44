#include "blocks.h"

namespace maa {

void conv_into(Ctx& ctx, const T4& x1, const T4* x2, const PackedW& w, const ConvOpt& o, T4& out) {
    IGemm p;
    p.a1 = x1.p;
    p.lda1 = x1.ld ? x1.ld : x1.C;
    p.C1 = x1.C;
    if (x2) {
        MAA_CHECK(x2->B == x1.B && x2->H == x1.H && x2->W == x1.W, "concat sources differ in shape");
        p.a2 = x2->p;
        p.lda2 = x2->C;
        p.C2 = x2->C;
    }
    p.Hin = x1.H;
    p.Win = x1.W;
    p.Hout = out.H;
    p.Wout = out.W;
    p.KH = o.KH;
    p.KW = o.KW;
    p.sh = p.sw = o.stride;
    p.dh = p.dw = o.dil;
    p.pw = o.pad;
    p.ph = o.pad_h >= 0 ? o.pad_h : (o.KH > 1 ? o.pad : 0);
    p.up = o.up;
    p.a_act = o.a_act;
    p.a_slope = o.a_slope;
    p.b = w.w;
    p.ldb = w.ld;
    p.b_nk = w.nk;
    p.b_split = w.split;
    if (x1.split) {
        MAA_CHECK(!x2, "a split source cannot be concatenated");
        p.a_split = 1;
    }
    p.M = out.B * out.H * out.W;
    p.K = o.KH * o.KW * (p.C1 + p.C2);
    MAA_CHECK(p.K == w.K, "conv weight K mismatch");
    p.N = o.geglu ? w.N : out.C;
    MAA_CHECK(p.N <= w.N, "conv weight N mismatch");
    p.bias = w.bias;
    p.rowadd = o.rowadd;
    p.ld_rowadd = o.ld_rowadd;
    p.res = o.res;
    p.ldr = out.C;
    p.act = o.act;
    p.act_slope = o.act_slope;
    p.out_scale = o.out_scale;
    p.accumulate = o.accumulate;
    p.geglu = o.geglu;
    p.c_split = o.c_split;
    p.c = out.p;
    p.ldc = out.C;
    p.c2 = o.c2;
    p.ldc2 = out.C;
    p.c2_slope = o.c2_slope;
    launch_igemm(ctx, p);
}

bool conv_up2_into(Ctx& ctx, const T4& x, const PackedW& w4, T4& out) {
    if (!ctx.tune.up2 || !w4.w || w4.phase_rows == 0 || ctx.dtype == 0 || x.split || x.C % 32 != 0 || x.ld != 0) return false;
    MAA_CHECK(out.B == x.B && out.H == 2 * x.H && out.W == 2 * x.W && out.C == w4.N, "conv_up2: output shape");
    const size_t mk = ctx.ws.mark();
    const long long M = x.rows();
    float* xs = ctx.ws.alloc_f((size_t)x.numel());                  // the source as split32 rows (8 MB at the UNet's 5 x 39 level)
    float* planes = ctx.ws.alloc_f((size_t)4 * M * out.C);          // [4 phases][B, H, W, C]
    IGemm p;
    p.a1 = xs;
    p.lda1 = x.C;
    p.C1 = x.C;
    p.a_split = 1;
    p.Hin = p.Hout = x.H;
    p.Win = p.Wout = x.W;
    p.KH = p.KW = 2;
    p.ph = p.pw = 1;
    p.b = w4.w;
    p.ldb = w4.ld;
    p.b_nk = w4.nk;
    p.b_split = w4.split;
    p.M = (int)M;
    p.K = 4 * x.C;
    p.N = out.C;
    p.bias = w4.bias;
    p.c = planes;
    p.ldc = out.C;
    p.ldr = out.C;
    p.ldc2 = out.C;
    p.zeros = ctx.zeros;
    MAA_CHECK(p.K == w4.K, "conv_up2 weight K mismatch");
    if (!ctx.ws.dry) launch_split32_pack(ctx, x.p, M, x.C, xs);
    const bool ok = launch_igemm_pp_up2(ctx, p, w4.Npad, (long long)w4.phase_rows * w4.ld);
    if (ok) launch_pixel_shuffle2(ctx, planes, x.B, x.H, x.W, out.C, out.p);
    ctx.ws.release(mk);
    return ok;
}

void linear_into(Ctx& ctx, const float* a, int lda, long long rows, int K, const PackedW& w, const float* res,
                 int ldr, float* out, int ldc, int geglu, int a_act, long long a_split_rows, int c_split, int act) {
    IGemm p;
    p.a1 = a;
    p.lda1 = lda;
    p.C1 = K;
    p.Hout = 1;
    p.Wout = 1;
    p.M = (int)rows;
    p.K = K;
    MAA_CHECK(K == w.K, "linear weight K mismatch");
    p.N = w.N;
    p.b = w.w;
    p.ldb = w.ld;
    p.b_nk = w.nk;
    p.b_split = w.split;
    if (a_split_rows > 0) {
        p.a_split = 1;
    }
    p.bias = w.bias;
    p.res = res;
    p.ldr = ldr;
    p.geglu = geglu;
    p.a_act = a_act;
    p.c_split = c_split;
    p.act = act;
    p.c = out;
    p.ldc = ldc;
    launch_igemm(ctx, p);
}

void attention_into(Ctx& ctx, const float* q, int ldq, int hsq, const float* k, int ldk, int hsk, const float* v,
                    int ldv, int hsv, int B, int heads, int dh, int Nq, int Nk, float alpha, float* out, int ldo,
                    int out_split, int causal) {
    MAA_CHECK(!causal || Nq == Nk, "causal attention needs Nq == Nk");
    if (launch_flash_attention(ctx, q, ldq, hsq, k, ldk, hsk, v, ldv, hsv, B, heads, dh, Nq, Nk, alpha, out, ldo,
                               out_split, causal))
        return;
    MAA_CHECK(!out_split, "split32 attention output needs the fused kernel");
    const size_t mk = ctx.ws.mark();
    const int ldS = (Nk + 3) / 4 * 4;
    float* S = ctx.ws.alloc_f((size_t)B * heads * Nq * ldS);
    IGemm p;
    p.a1 = q;
    p.lda1 = ldq;
    p.C1 = dh;
    p.M = Nq;
    p.K = dh;
    p.N = Nk;
    p.b = k;
    p.ldb = ldk;
    p.b_nk = 1;
    p.Z = B * heads;
    p.zin = heads;
    p.a_so = (long long)Nq * ldq;
    p.a_si = hsq;
    p.b_so = (long long)Nk * ldk;
    p.b_si = hsk;
    p.c_so = (long long)heads * Nq * ldS;
    p.c_si = (long long)Nq * ldS;
    p.alpha = alpha;
    p.c = S;
    p.ldc = ldS;
    launch_igemm(ctx, p);
    launch_softmax(ctx, S, (long long)B * heads * Nq, Nk, ldS, causal ? Nq : 0);
    IGemm r;
    r.a1 = S;
    r.lda1 = ldS;
    r.C1 = Nk;
    r.M = Nq;
    r.K = Nk;
    r.N = dh;
    r.b = v;
    r.ldb = ldv;
    r.Z = B * heads;
    r.zin = heads;
    r.a_so = (long long)heads * Nq * ldS;
    r.a_si = (long long)Nq * ldS;
    r.b_so = (long long)Nk * ldv;
    r.b_si = hsv;
    r.c_so = (long long)Nq * ldo;
    r.c_si = dh;
    r.c = out;
    r.ldc = ldo;
    launch_igemm(ctx, r);
    ctx.ws.release(mk);
}

}  // namespace maa
