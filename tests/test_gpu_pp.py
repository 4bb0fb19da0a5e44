"""The halo-staged ping-pong engine (csrc/igemm_pp.hip) at the operator level: 3x3 / stride 1 / "same" convolutions with the
activation handed over pre-split (MAA_OP_PRESPLIT=1, the form GroupNorm writes inside the models), every (tile width, K
slices) instantiation forced with MAA_PP, against torch.nn.functional.conv2d in fp32 on the CPU at the bf16x3 operator
tolerance (rel-max 2e-4).  Plus the engine's contracts: repeated runs are bit-identical and a sample's rows do not depend on
the batch they were computed in (the tile width and the number of K slices are functions of the layer only).
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from tests.util import check

pytestmark = pytest.mark.gpu

TOL = 2e-4
VARIANTS = ["128,1", "128,2", "128,3", "128,5", "160,1", "160,2", "160,4"]      # tile width, K slices


@pytest.fixture(scope="module")
def ctx():
    from audiogpt_amd.backend import Context
    c = Context("cuda:0", precision="bf16x3")
    yield c
    c.close()


class forced:
    """Environment for one call: the library parses the MAA_* knobs when a context is created: reload_tuning() re-reads them."""

    def __init__(self, pp, presplit=True, pp1=None, dma2=None):
        self.env = {"MAA_PP": pp, "MAA_OP_PRESPLIT": "1" if presplit else "0"}
        if pp1 is not None:
            self.env["MAA_PP1"] = pp1
        if dma2 is not None:
            self.env["MAA_DMA2"] = dma2

    def __enter__(self):
        from audiogpt_amd.backend import reload_tuning
        self.saved = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        reload_tuning()

    def __exit__(self, *a):
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        from audiogpt_amd.backend import reload_tuning
        reload_tuning()


def g(seed):
    return torch.Generator().manual_seed(seed)


CONVS = [  # B, Cin, Cout, H, W
    (2, 320, 320, 10, 78),       # M = 1560: 6.1 tiles of 256, N = 2.5 / 2 tiles, 10 channel chunks
    (3, 640, 640, 5, 39),        # a tile spans more than one sample (195 positions each): every edge mask in play
    (1, 96, 200, 5, 39),         # M = 195 < one tile, N not a multiple of 32, 3 channel chunks (fewer than some slice counts)
    (2, 64, 96, 7, 9),           # tiny image: the halo of a tile covers several samples
    (1, 32, 64, 3, 100),         # widest image the two A buffers still fit (W = 100 with 128-wide tiles)
]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("B,Cin,Cout,H,W", CONVS)
def test_conv3x3(ctx, variant, B, Cin, Cout, H, W):
    x = torch.randn(B, Cin, H, W, generator=g(7))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(8)) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g(9))
    with forced(variant):
        y = ctx.op_conv(x, w, b, pad=1)
    check(f"pp[{variant}]_conv3x3_{Cin}_{Cout}_{H}x{W}_b{B}", y, F.conv2d(x, w, b, padding=1), TOL)


def test_position_and_channel_probe(ctx):
    """One-hot weights: output channel n copies input channel n of ONE tap -- a misplaced tap offset, edge mask, channel
    chunk or weight row shows up as a wrong or shifted plane rather than as noise."""
    B, C, H, W = 2, 64, 6, 11
    x = torch.randn(B, C, H, W, generator=g(3))
    for t in range(9):
        w = torch.zeros(C, C, 3, 3)
        w[torch.arange(C), torch.arange(C), t // 3, t % 3] = 1.0
        for variant in ("128,1", "128,2"):
            with forced(variant):
                y = ctx.op_conv(x, w, None, pad=1)
            ref = F.conv2d(x, w, None, padding=1)
            assert (y.cpu() - ref).abs().max() < 1e-4, (t, variant, float((y.cpu() - ref).abs().max()))      # x is rebuilt from hi + lo


@pytest.mark.parametrize("variant", ["128,1", "128,3", "160,2", None])
def test_deterministic_and_batch_invariant(ctx, variant):
    """None = the default policy."""
    x = torch.randn(6, 640, 5, 39, generator=g(21))
    w = torch.randn(640, 640, 3, 3, generator=g(22)) / math.sqrt(5760)
    b = torch.randn(640, generator=g(23))
    with forced(variant if variant else ""):
        y6 = ctx.op_conv(x, w, b, pad=1).cpu()
        y6b = ctx.op_conv(x, w, b, pad=1).cpu()
        y1 = ctx.op_conv(x[4:5], w, b, pad=1).cpu()
    assert torch.equal(y6, y6b)
    assert torch.equal(y6[4:5], y1)
    check(f"pp[{variant}]_conv_640_b6", y6, F.conv2d(x, w, b, padding=1), TOL)


def test_engine_is_selected_by_default_and_can_be_switched_off(ctx):
    """The default policy takes the UNet's 3x3 convolutions; MAA_PP=off hands them back to the second engine -- both meet the
    tolerance (they differ in the last bits: different order of the k-steps)."""
    x = torch.randn(2, 320, 10, 78, generator=g(31))
    w = torch.randn(320, 320, 3, 3, generator=g(32)) / math.sqrt(2880)
    ref = F.conv2d(x, w, None, padding=1)
    ctx.prof_begin(detail=True)
    with forced(""):
        y = ctx.op_conv(x, w, None, pad=1)
    rows = ctx.prof_end()
    assert any(k.startswith("pp") for k in rows), rows.keys()
    ctx.prof_begin(detail=True)
    with forced("off"):
        y_off = ctx.op_conv(x, w, None, pad=1)
    rows = ctx.prof_end()
    assert not any(k.startswith("pp") for k in rows), rows.keys()
    check("pp_default_conv_320", y, ref, TOL)
    check("pp_off_conv_320", y_off, ref, TOL)


# ---------------------------------------------------------------------------------------------- the 1x1 / Linear form
VARIANTS1 = ["128,1", "128,2", "128,3", "160,1", "160,2"]


@pytest.mark.parametrize("variant", VARIANTS1)
@pytest.mark.parametrize("M,K,N,bias", [(1560, 320, 320, True), (390, 640, 640, False), (130, 2560, 640, True),
                                         (257, 64, 77, False), (3120, 640, 1920, True)])
def test_linear(ctx, variant, M, K, N, bias):
    a = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3)) if bias else None
    with forced("off", pp1=variant):
        y = ctx.op_linear(a, w, b)
    check(f"pp1[{variant}]_linear_{M}x{K}x{N}", y, F.linear(a, w, b), TOL)


@pytest.mark.parametrize("variant", ["128,1", "128,2"])
def test_linear_geglu(ctx, variant):
    a = torch.randn(1560, 320, generator=g(4))
    w = torch.randn(2560, 320, generator=g(5)) / math.sqrt(320)
    b = torch.randn(2560, generator=g(6)) * 0.1
    val, gate = F.linear(a, w, b).chunk(2, dim=-1)
    with forced("off", pp1=variant):
        y = ctx.op_linear(a, w, b, geglu=True)
    check(f"pp1[{variant}]_geglu", y, val * F.gelu(gate), TOL)


def test_identity_asymmetric(ctx):
    """A = I against an asymmetric B: catches a transposed or mis-placed output block (fragment map, slab layout)."""
    K = N = 256
    a = torch.eye(K)
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 17.0 + torch.arange(N)[:, None] * 0.5
    for variant in VARIANTS1:
        with forced("off", pp1=variant):
            y = ctx.op_linear(a, w)
        check(f"pp1[{variant}]_identity", y, w.t().contiguous(), TOL)


def test_linear_form_is_bit_identical_to_the_other_engines(ctx):
    """Same products in the same order per accumulator: without a K split the 1x1 form equals the 64x64 LDS-DMA engine bit
    for bit, with two slices the second engine's two-slice result."""
    a = torch.randn(3120, 640, generator=g(11))
    w = torch.randn(640, 640, generator=g(12)) / math.sqrt(640)
    b = torch.randn(640, generator=g(13))
    with forced("off", pp1="off", dma2="off"):
        y_dma = ctx.op_linear(a, w, b).cpu()
    for variant in ("128,1", "160,1"):
        with forced("off", pp1=variant):
            assert torch.equal(ctx.op_linear(a, w, b).cpu(), y_dma), variant
    with forced("off", pp1="off", dma2="0,4,1,2"):
        y2 = ctx.op_linear(a, w, b).cpu()
    with forced("off", pp1="128,2"):
        assert torch.equal(ctx.op_linear(a, w, b).cpu(), y2)


# ---------------------------------------------------------------------------------------------- dilated 1-D kernels (vocoders)
@pytest.mark.parametrize("k,dil", [(3, 1), (3, 5), (7, 1), (7, 3), (11, 1), (11, 5)])
@pytest.mark.parametrize("B,C,Cout,L", [(2, 128, 128, 1000), (3, 256, 256, 300), (1, 512, 512, 77)])
def test_conv1d_dilated(ctx, k, dil, B, C, Cout, L):
    """The MRF convolutions of the wide HiFi-GAN stages: "same" padding, 3 / 7 / 11 taps, dilation 1 / 3 / 5; L = 300 and
    77 put several samples into one 256-row tile (every edge mask in play); the 3-tap kernels take the instantiation that
    issues three A pieces per memory phase."""
    x = torch.randn(B, C, 1, L, generator=g(51))
    w = torch.randn(Cout, C, 1, k, generator=g(52)) / math.sqrt(k * C)
    b = torch.randn(Cout, generator=g(53))
    pad = dil * (k - 1) // 2
    ctx.prof_begin(detail=True)
    with forced(""):
        y = ctx.op_conv(x, w, b, pad=pad, dil=dil)
    rows = ctx.prof_end()
    assert any(r.startswith("pp") for r in rows), rows.keys()
    check(f"pp_conv1d_k{k}_d{dil}_{C}_{Cout}_L{L}_b{B}", y, F.conv2d(x, w, b, padding=(0, pad), dilation=(1, dil)), TOL)


UP2 = [  # B, Cin, Cout, H, W  (low-resolution source)
    (16, 640, 640, 5, 39),     # the T2A UNet's Upsample at the benchmark's batch: 4 phases x 52 tiles of 256 x 160
    (2, 640, 640, 5, 39),      # the golden's batch: ragged last M tile (390 rows)
    (3, 320, 320, 10, 53),     # inpaint-like width, odd batch, two N tiles
    (1, 512, 512, 10, 78),     # the VAE decoder's first Upsample
    (2, 256, 256, 20, 156),    # ... its second: 128-wide tiles, persistent workgroups (items > CUs)
    (2, 128, 128, 40, 312),    # ... its last: a chunk of 256 + W + 1 lines does not fit the A ring twice over -> the gather path (asserted)
    (1, 96, 64, 7, 9),         # N = 64 (one padded 128-wide tile), three channel chunks, tiny image
]


@pytest.mark.parametrize("case", UP2, ids=lambda c: "x".join(map(str, c)))
def test_upsample_conv_as_four_phase_convolutions(ctx, case):
    """Upsample (nearest 2x) + conv3x3 (openaimodel.py:116-118; model.py:52-56) in the bf16 modes runs as four 2x2 convolutions of the
    LOW-resolution source, one per parity of the output pixel, the 3x3 taps that read the same source pixel summed at load (4 / 9 of the
    multiplications), in ONE launch of the ping-pong engine (UP2 instantiation) + a pixel shuffle: against
    conv2d(interpolate(x, 2, "nearest")) in fp32 on the CPU, image borders and sample boundaries included; the kernel really ran;
    a sample's rows do not depend on its batch; with MAA_PP=off the old gather path gives the same answer to tolerance."""
    B, Cin, Cout, H, W = case
    x = torch.randn(B, Cin, H, W, generator=g(B * 7 + W))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(Cin + Cout)) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g(5))
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    ctx.prof_begin()
    y = ctx.op_conv(x, w, b, pad=1, up=True).cpu()
    rows = ctx.prof_end()
    if W >= 300:      # the phase form is not taken (LDS): same call, same answer, through the virtual-upsample gather
        assert not any(k.startswith("igemm_pp_up2") for k in rows), rows.keys()
        check(f"pp_up2_fallback_{'x'.join(map(str, case))}", y, ref, TOL)
        return
    assert any(k.startswith("igemm_pp_up2_bf16x3") for k in rows) and "pixel_shuffle2_kernel" in rows, rows.keys()
    assert y.shape == ref.shape == (B, Cout, 2 * H, 2 * W)
    check(f"pp_up2_{'x'.join(map(str, case))}", y, ref, TOL)
    y1 = ctx.op_conv(x[:1], w, b, pad=1, up=True).cpu()
    assert torch.equal(y1, y[:1])                                   # batch invariance, bit for bit
    assert torch.equal(ctx.op_conv(x, w, b, pad=1, up=True).cpu(), y)      # deterministic
    with forced("off", presplit=False):
        ctx.prof_begin()
        y_old = ctx.op_conv(x, w, b, pad=1, up=True).cpu()
        assert not any(k.startswith("igemm_pp_up2") for k in ctx.prof_end())
    check(f"pp_up2_vs_gather_{'x'.join(map(str, case))}", y, y_old, TOL)


@pytest.mark.parametrize("case", [(16, 320, 4, 10, 78), (3, 320, 4, 10, 106), (1, 96, 3, 5, 7), (2, 640, 4, 3, 5)], ids=lambda c: "x".join(map(str, c)))
def test_narrow_output_convolution(ctx, case):
    """The UNet's last layer, conv3x3 320 -> 4 (openaimodel.py:693-697), on its own kernel in the bf16x3 mode (misc.hip
    narrow_conv3x3_kernel: split32 rows in, fp32 FMAs, NCHW out) against conv2d in fp32 on the CPU: borders, sample boundaries, a
    position count that is not a multiple of 16, channel counts beyond one pass of a wave (640) and N = 3."""
    B, Cin, Cout, H, W = case
    x = torch.randn(B, Cin, H, W, generator=g(W + Cin))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(Cin)) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g(9))
    with forced("", presplit=True):
        ctx.prof_begin()
        y = ctx.op_conv(x, w, b, pad=1).cpu()
        rows = ctx.prof_end()
        y1 = ctx.op_conv(x[:1], w, b, pad=1).cpu()
    assert "narrow_conv3x3_kernel" in rows, rows.keys()
    check(f"narrow_conv_{'x'.join(map(str, case))}", y, F.conv2d(x, w, b, padding=1), TOL)
    assert torch.equal(y1, y[:1])
