"""Operator-level parity of the HIP kernels (through the C ABI) against plain PyTorch fp32 on CPU.

Tolerances are fp32 reduction-order noise: rel-max 2e-5 for K up to a few thousand."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import check

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def ctx():
    from audiogpt_amd.backend import Context
    c = Context("cuda:0")
    yield c
    c.close()


def g(seed):
    return torch.Generator().manual_seed(seed)


def test_mfma_layout_identity_asymmetric(ctx):
    """A = I against an asymmetric B catches any row/column swap of the MFMA fragment maps."""
    K = N = 96
    a = torch.eye(K)
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 17.0 + torch.arange(N)[:, None] * 0.5
    y = ctx.op_linear(a, w)
    check("mfma_identity", y, w.t().contiguous(), 1e-6)


@pytest.mark.parametrize("M,K,N,bias", [(1560, 320, 320, True), (16, 1280, 6080, True), (390, 640, 640, False),
                                         (3, 320, 1280, True), (1000, 1024, 640, False), (257, 40, 77, False),
                                         (130, 2560, 640, True), (65, 32, 33, True)])
def test_linear(ctx, M, K, N, bias):
    a = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3)) if bias else None
    y = ctx.op_linear(a, w, b)
    check(f"linear_{M}x{K}x{N}", y, F.linear(a, w, b), TOL)


@pytest.mark.parametrize("M,K,inner", [(1560, 320, 1280), (390, 640, 2560), (100, 64, 32)])
def test_linear_geglu(ctx, M, K, inner):
    a = torch.randn(M, K, generator=g(4))
    w = torch.randn(2 * inner, K, generator=g(5)) / math.sqrt(K)
    b = torch.randn(2 * inner, generator=g(6)) * 0.1
    y = ctx.op_linear(a, w, b, geglu=True)
    val, gate = F.linear(a, w, b).chunk(2, dim=-1)
    check(f"geglu_{M}x{K}x{inner}", y, val * F.gelu(gate), TOL)


@pytest.mark.parametrize("B,Cin,Cout,H,W,stride,up", [
    (2, 320, 320, 10, 78, 1, False), (2, 320, 320, 10, 78, 2, False), (2, 640, 640, 5, 39, 1, True),
    (2, 4, 320, 10, 78, 1, False), (1, 9, 320, 10, 106, 1, False), (2, 320, 4, 10, 78, 1, False),
    (1, 128, 1, 16, 40, 1, False), (1, 1, 128, 16, 24, 1, False), (3, 64, 96, 7, 9, 1, False)])
def test_conv3x3(ctx, B, Cin, Cout, H, W, stride, up):
    x = torch.randn(B, Cin, H, W, generator=g(7))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(8)) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g(9))
    y = ctx.op_conv(x, w, b, stride=stride, pad=1, up=up)
    xr = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    check(f"conv3x3_{Cin}_{Cout}_{H}x{W}_s{stride}_up{int(up)}", y, F.conv2d(xr, w, b, stride=stride, padding=1), TOL)


def test_conv3x3_vae_downsample_padding(ctx):
    """VAE Downsample: pad right/bottom only, stride 2, no left/top pad (model.py:72-77)."""
    x = torch.randn(1, 32, 12, 20, generator=g(10))
    w = torch.randn(32, 32, 3, 3, generator=g(11)) / 17.0
    b = torch.randn(32, generator=g(12))
    y = ctx.op_conv(x, w, b, stride=2, pad=0, out_hw=(6, 10))
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    check("conv_vae_down", y, ref, TOL)


def test_conv1x1(ctx):
    x = torch.randn(2, 960, 5, 39, generator=g(13))
    w = torch.randn(640, 960, 1, 1, generator=g(14)) / 31.0
    b = torch.randn(640, generator=g(15))
    check("conv1x1", ctx.op_conv(x, w, b), F.conv2d(x, w, b), TOL)


@pytest.mark.parametrize("C,k,d,L,B,leaky", [(256, 3, 1, 512, 2, 0.1), (128, 7, 3, 1000, 1, 0.1), (64, 11, 5, 2048, 2, 0.1),
                                             (32, 11, 1, 4096, 1, 0.1), (32, 3, 5, 300, 3, 0.0)])
def test_conv1d_dilated(ctx, C, k, d, L, B, leaky):
    x = torch.randn(B, C, 1, L, generator=g(16))
    w = torch.randn(C, C, 1, k, generator=g(17)) / math.sqrt(C * k)
    b = torch.randn(C, generator=g(18))
    pad = (k * d - d) // 2
    y = ctx.op_conv(x, w, b, pad=pad, dil=d, leaky=leaky)
    xin = F.leaky_relu(x, leaky) if leaky else x
    ref = F.conv1d(xin[:, :, 0], w[:, :, 0], b, padding=pad, dilation=d)[:, :, None]
    check(f"conv1d_C{C}_k{k}_d{d}", y, ref, TOL)


def test_conv1d_pre_and_post(ctx):
    x = torch.randn(2, 80, 1, 200, generator=g(19))
    w = torch.randn(512, 80, 1, 7, generator=g(20)) / math.sqrt(560)
    b = torch.randn(512, generator=g(21))
    check("conv_pre", ctx.op_conv(x, w, b, pad=3), F.conv1d(x[:, :, 0], w[:, :, 0], b, padding=3)[:, :, None], TOL)
    x = torch.randn(2, 32, 1, 5000, generator=g(22))
    w = torch.randn(1, 32, 1, 7, generator=g(23)) / 15.0
    b = torch.randn(1, generator=g(24))
    y = ctx.op_conv(x, w, b, pad=3, leaky=0.01)
    ref = F.conv1d(F.leaky_relu(x[:, :, 0], 0.01), w[:, :, 0], b, padding=3)[:, :, None]
    check("conv_post", y, ref, TOL)


@pytest.mark.parametrize("Cin,Cout,k,s,L", [(512, 256, 16, 8, 100), (128, 64, 4, 2, 999), (64, 32, 4, 2, 64), (96, 48, 16, 8, 33)])
def test_conv_transpose1d(ctx, Cin, Cout, k, s, L):
    x = torch.randn(2, Cin, L, generator=g(25))
    w = torch.randn(Cin, Cout, k, generator=g(26)) / math.sqrt(Cin * k / s)
    b = torch.randn(Cout, generator=g(27))
    y = ctx.op_conv_transpose1d(x, w, b, s, leaky=0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=(k - s) // 2)
    check(f"convtr_{Cin}_{Cout}_k{k}_s{s}", y, ref, TOL)


@pytest.mark.parametrize("C,HW,eps,silu", [(320, 780, 1e-5, True), (960, 195, 1e-5, True), (1280, 195, 1e-6, False),
                                           (128, 4096, 1e-6, True), (512, 780, 1e-6, False), (32, 50, 1e-5, True),
                                           (64, 1003, 1e-5, False), (1920, 780, 1e-5, True)])
def test_groupnorm(ctx, C, HW, eps, silu):
    x = torch.randn(2, C, HW, generator=g(28)) * 2.0 + 0.5
    ga = torch.randn(C, generator=g(29))
    be = torch.randn(C, generator=g(30))
    y = ctx.op_groupnorm(x, ga, be, eps, silu)
    ref = F.group_norm(x, 32, ga, be, eps)
    if silu:
        ref = F.silu(ref)
    check(f"groupnorm_C{C}_HW{HW}", y, ref, 2e-5)


@pytest.mark.parametrize("rows,C", [(1560, 320), (390, 640), (7, 1024), (5, 256)])
def test_layernorm(ctx, rows, C):
    x = torch.randn(rows, C, generator=g(31)) * 3 + 1
    ga = torch.randn(C, generator=g(32))
    be = torch.randn(C, generator=g(33))
    check(f"layernorm_{rows}x{C}", ctx.op_layernorm(x, ga, be), F.layer_norm(x, (C,), ga, be, 1e-5), 1e-5)


@pytest.mark.parametrize("B,heads,dh,Nq,Nk", [(2, 8, 40, 780, 780), (2, 8, 80, 195, 195), (2, 8, 40, 780, 77),
                                              (2, 16, 32, 195, 1), (1, 1, 512, 780, 780), (1, 1, 256, 1200, 1200),
                                              (1, 8, 40, 1060, 1060)])
def test_attention(ctx, B, heads, dh, Nq, Nk):
    C = heads * dh
    q = torch.randn(B, Nq, C, generator=g(34))
    k = torch.randn(B, Nk, C, generator=g(35))
    v = torch.randn(B, Nk, C, generator=g(36))
    alpha = dh ** -0.5
    y = ctx.op_attention(q, k, v, heads, alpha)

    def split(t):
        return t.reshape(B, t.shape[1], heads, dh).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bhjd->bhij", split(q), split(k)) * alpha
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), split(v)).permute(0, 2, 1, 3).reshape(B, Nq, C)
    check(f"attention_h{heads}_d{dh}_{Nq}x{Nk}", y, ref, 2e-5)


def test_snake_antialiased_activation(ctx):
    from oracle import vocoder as O
    x = torch.randn(2, 48, 333, generator=g(37))
    al = torch.randn(48, generator=g(38)) * 0.3
    be = torch.randn(48, generator=g(39)) * 0.3
    filt = O.kaiser_sinc_filter1d(0.25, 0.3, 12)
    ref = O.downsample1d(O.snake(O.upsample1d(x, filt), al, be, True), filt)
    check("snake_aa", ctx.op_snake_aa(x, al, be, True), ref, 1e-5)


@pytest.mark.parametrize("C,HW", [(320, 780), (640, 780), (960, 780), (640, 195), (1280, 195), (1920, 195), (256, 780), (512, 3120)])
def test_groupnorm_one_pass_against_two_pass_and_batch_invariance(ctx, C, HW):
    """Round 5: GroupNorm reads the tensor once (a workgroup keeps a sample's rows of a few groups in registers: norm.hip
    gn_fused_kernel) where the registers hold them; MAA_GN_TWO_PASS=1 keeps the statistics + apply launches.  Both against torch,
    the two against each other, and a sample alone against the same sample inside a batch (bit for bit: the layout is a function
    of the layer, a workgroup sees one sample)."""
    import os
    from audiogpt_amd.backend import reload_tuning
    x = torch.randn(5, C, HW, generator=g(128)) * 1.7 + 0.3
    ga, be = torch.randn(C, generator=g(129)), torch.randn(C, generator=g(130))
    ref = F.silu(F.group_norm(x, 32, ga, be, 1e-5))
    y1 = ctx.op_groupnorm(x, ga, be, 1e-5, True).cpu()
    y1b = ctx.op_groupnorm(x, ga, be, 1e-5, True).cpu()
    one = ctx.op_groupnorm(x[3:4], ga, be, 1e-5, True).cpu()
    os.environ["MAA_GN_TWO_PASS"] = "1"
    try:
        reload_tuning()
        y2 = ctx.op_groupnorm(x, ga, be, 1e-5, True).cpu()
    finally:
        os.environ.pop("MAA_GN_TWO_PASS")
        reload_tuning()
    check(f"groupnorm_onepass_C{C}_HW{HW}", y1, ref, 2e-5)
    check(f"groupnorm_twopass_C{C}_HW{HW}", y2, ref, 2e-5)
    assert torch.equal(y1, y1b) and torch.equal(one, y1[3:4])
    assert float((y1 - y2).abs().max()) <= 2e-5 * float(ref.abs().max())
