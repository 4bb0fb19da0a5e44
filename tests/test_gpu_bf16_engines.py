"""The plain-bf16 mode (precision="bf16", BASELINE configs[1]'s literal dtype) on the engines that ship: the one-product
(TERMS = 1) instantiations of the LDS-DMA engines -- igemm_dma (64x64 .. 128x128 tiles), igemm_dma2 (split-K), igemm_pp (the
halo-staged 3x3 / dilated 1-D form) and igemm_pp1 (its 1x1 form) -- whose operands are the hi halves of the same split32 lines
the bf16x3 mode reads (hi = bf16(x), round to nearest even: exactly the rounding the mode asks for).

The reference of every case is the SAME contraction on operands rounded to bf16 (fp64 accumulation on the CPU): what is left is
the fp32 summation order, so the tolerance is the bf16x3 operator tolerance (rel-max 2e-5 would do; 1e-4 is asserted), not the
loose 5e-2 of "bf16 against fp32".  Engines that keep the k order of the register-staged bf16 engine must equal it bit for bit.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from tests.util import check

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    from audiogpt_amd.backend import Context
    c = Context("cuda:0", precision="bf16")
    yield c
    c.close()


class forced:
    """The library parses the MAA_* knobs when a context is created; reload_tuning() re-reads them."""

    def __init__(self, **env):
        self.env = {"MAA_OP_PRESPLIT": "1"}
        self.env.update({k: v for k, v in env.items() if v is not None})

    def __enter__(self):
        from audiogpt_amd.backend import reload_tuning
        self.saved = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        reload_tuning()

    def __exit__(self, *a):
        from audiogpt_amd.backend import reload_tuning
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        reload_tuning()


def g(seed):
    return torch.Generator().manual_seed(seed)


def r16(t):
    """The operand the mode multiplies: rounded to bf16 (round to nearest even), as float64 for the reference."""
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("variant", [None, "128,1", "128,2", "160,1", "160,2", "160,4"])
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 320, 320, 10, 78), (3, 640, 640, 5, 39), (1, 96, 200, 5, 39), (2, 64, 96, 7, 9)])
def test_conv3x3_on_the_ping_pong_engine(ctx, variant, B, Cin, Cout, H, W):
    x = torch.randn(B, Cin, H, W, generator=g(7))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(8)) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g(9))
    ctx.prof_begin(detail=True)
    with forced(MAA_PP=variant or ""):
        y = ctx.op_conv(x, w, b, pad=1)
    rows = ctx.prof_end()
    assert any(k.startswith("pp") for k in rows), rows.keys()
    ref = F.conv2d(r16(x), r16(w), b.double(), padding=1)
    check(f"bf16_pp[{variant}]_conv3x3_{Cin}_{Cout}_{H}x{W}_b{B}", y, ref, TOL)


@pytest.mark.parametrize("k,dil", [(3, 1), (7, 3), (11, 5)])
def test_conv1d_dilated_on_the_ping_pong_engine(ctx, k, dil):
    B, C, L = 2, 256, 300
    x = torch.randn(B, C, 1, L, generator=g(51))
    w = torch.randn(C, C, 1, k, generator=g(52)) / math.sqrt(k * C)
    b = torch.randn(C, generator=g(53))
    pad = dil * (k - 1) // 2
    ctx.prof_begin(detail=True)
    with forced(MAA_PP=""):
        y = ctx.op_conv(x, w, b, pad=pad, dil=dil)
    rows = ctx.prof_end()
    assert any(r.startswith("pp") for r in rows), rows.keys()
    ref = F.conv2d(r16(x), r16(w), b.double(), padding=(0, pad), dilation=(1, dil))
    check(f"bf16_pp_conv1d_k{k}_d{dil}", y, ref, TOL)


LINEARS = [(1560, 320, 320, True), (390, 640, 640, False), (130, 2560, 640, True), (257, 64, 77, False), (3120, 640, 1920, True)]


@pytest.mark.parametrize("engine", ["dma", "pp1_128", "pp1_160", "dma2_s1", "dma2_s2"])
@pytest.mark.parametrize("M,K,N,bias", LINEARS)
def test_linear(ctx, engine, M, K, N, bias):
    a = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3)) if bias else None
    env = {"dma": dict(MAA_PP1="off", MAA_DMA2="off"), "pp1_128": dict(MAA_PP1="128,1"), "pp1_160": dict(MAA_PP1="160,1"),
           "dma2_s1": dict(MAA_PP1="off", MAA_DMA2="0,4,1,1"), "dma2_s2": dict(MAA_PP1="off", MAA_DMA2="0,4,1,2")}[engine]
    with forced(MAA_PP="off", **env):
        y = ctx.op_linear(a, w, b)
    ref = F.linear(r16(a), r16(w), b.double() if bias else None)
    check(f"bf16_{engine}_linear_{M}x{K}x{N}", y, ref, TOL)


def test_linear_geglu(ctx):
    a = torch.randn(1560, 320, generator=g(4))
    w = torch.randn(2560, 320, generator=g(5)) / math.sqrt(320)
    b = torch.randn(2560, generator=g(6)) * 0.1
    val, gate = F.linear(r16(a), r16(w), b.double()).chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    for tag, env in (("dma", dict(MAA_PP1="off")), ("pp1", dict(MAA_PP1="128,1"))):
        with forced(MAA_PP="off", MAA_DMA2="off", **env):
            y = ctx.op_linear(a, w, b, geglu=True)
        check(f"bf16_{tag}_geglu", y, ref, TOL)


def test_engines_with_the_register_engines_k_order_are_bit_identical_to_it(ctx):
    """One product per k-step, k ascending, fp32 accumulation: the LDS-DMA engine, the second engine without a K split and the 1x1
    ping-pong form must return the register-staged bf16 engine's bits (MAA_NO_DMA=1); two K slices those of the second engine."""
    a = torch.randn(3120, 640, generator=g(11))
    w = torch.randn(640, 640, generator=g(12)) / math.sqrt(640)
    b = torch.randn(640, generator=g(13))
    with forced(MAA_NO_DMA="1"):
        y_reg = ctx.op_linear(a, w, b).cpu()
    for tag, env in (("dma", dict(MAA_PP1="off", MAA_DMA2="off")), ("dma2", dict(MAA_PP1="off", MAA_DMA2="0,4,1,1")),
                     ("pp1_128", dict(MAA_PP1="128,1")), ("pp1_160", dict(MAA_PP1="160,1"))):
        with forced(MAA_PP="off", **env):
            assert torch.equal(ctx.op_linear(a, w, b).cpu(), y_reg), tag
    with forced(MAA_PP="off", MAA_PP1="off", MAA_DMA2="0,4,1,2"):
        y2 = ctx.op_linear(a, w, b).cpu()
    with forced(MAA_PP="off", MAA_PP1="128,2"):
        assert torch.equal(ctx.op_linear(a, w, b).cpu(), y2)
    # a 3x3 convolution: the LDS-DMA engine against the register engine (same (tap, channel) k order)
    x = torch.randn(2, 320, 10, 78, generator=g(21))
    wc = torch.randn(320, 320, 3, 3, generator=g(22)) / math.sqrt(2880)
    with forced(MAA_NO_DMA="1"):
        c_reg = ctx.op_conv(x, wc, None, pad=1).cpu()
    with forced(MAA_PP="off", MAA_DMA2="off"):
        assert torch.equal(ctx.op_conv(x, wc, None, pad=1).cpu(), c_reg)


def test_the_unet_takes_the_shipped_engines_and_is_batch_invariant(golden, ctx):
    """The whole T2A UNet in the plain-bf16 mode: its contractions run on the engines the bf16x3 mode ships (none on the
    register-staged kernel except the shapes that kernel also takes in bf16x3), the error against the reference stays at the
    bf16 level, and a sample's result does not depend on its batch."""
    from audiogpt_amd import config as C
    from audiogpt_amd import weights as WT
    from audiogpt_amd.backend import UNet
    from tests.util import rel_err
    gu = golden("unet_t2a")
    unet = UNet(ctx, C.UNET_T2A, WT.make_unet_state_dict(C.UNET_T2A, seed=0))
    x, t, c = torch.from_numpy(gu["x"]), torch.from_numpy(gu["t"]), torch.from_numpy(gu["context"])
    ctx.prof_begin()
    y = unet(x, t, c).cpu()
    rows = ctx.prof_end()
    yb = unet(torch.cat([x, x.flip(0), x]), torch.cat([t, t.flip(0), t]), torch.cat([c, c.flip(0), c])).cpu()
    y1 = unet(x[1:2], t[1:2], c[1:2]).cpu()
    unet.close()
    names = set(rows)
    assert any(n.startswith("igemm_pp_bf16<") for n in names), names
    assert any(n.startswith("igemm_dma_bf16<") for n in names), names
    assert not any("bf16x3" in n for n in names), names
    assert "flash_attention" in names, names
    r, _, _ = rel_err(y, gu["y"])
    assert r <= 5e-2, r
    assert torch.equal(yb[1:2], y1) and torch.equal(yb[2:3], y1)
