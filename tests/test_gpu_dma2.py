"""The wide-tile / split-K LDS-DMA engine (csrc/igemm_dma2.hip) at the operator level.

The engine keeps one instantiation since round 3 (128x128 tiles, four stages, pipelined loop; the 3x3 convolutions moved to
igemm_pp.hip, tests/test_gpu_pp.py); it is forced with 1 / 2 / 3 / 5 K slices onto every eligible problem (MAA_DMA2) with
the activation handed over pre-split (MAA_OP_PRESPLIT=1, the form the normalisations write inside the models) and
compared with torch.nn.functional in fp32 on the CPU at the bf16x3 operator tolerance (rel-max 2e-4).  Plus the
engine's two contracts: S = 1 is bit-identical to the other bf16x3 engines, and results do not depend on the batch.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from tests.util import check

pytestmark = pytest.mark.gpu

TOL = 2e-4
# tile (0: 128x128 / 4 waves), LDS stages, in-wave pipelining, K slices [, minimum K, maximum K]
VARIANTS = ["0,4,1,1", "0,4,1,2", "0,4,1,3", "0,4,1,5"]


@pytest.fixture(scope="module")
def ctx():
    from audiogpt_amd.backend import Context
    c = Context("cuda:0", precision="bf16x3")
    yield c
    c.close()


class forced:
    """Environment for one call: the library parses the MAA_* knobs when a context is created: reload_tuning() re-reads them."""

    def __init__(self, dma2, presplit=True):
        # (the ping-pong engines would take the 3x3 convolutions / some linears first: off)
        self.env = {"MAA_DMA2": dma2, "MAA_OP_PRESPLIT": "1" if presplit else "0", "MAA_PP": "off", "MAA_PP1": "off"}

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


CONVS = [  # B, Cin, Cout, H, W, stride, up
    (2, 320, 320, 10, 78, 1, False),      # M = 1560 (ragged against 128 and 256), K = 2880
    (2, 320, 320, 10, 78, 2, False),      # strided gather
    (2, 640, 640, 5, 39, 1, True),        # virtual nearest-2x upsample, K = 5760
    (3, 64, 96, 7, 9, 1, False),          # N < one tile, 18 chunks: fewer chunks than some slice counts ask for
    (1, 96, 200, 5, 39, 1, False),        # M = 195 < one tile, N not a multiple of 32
]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("B,Cin,Cout,H,W,stride,up", CONVS)
def test_conv3x3(ctx, variant, B, Cin, Cout, H, W, stride, up):
    x = torch.randn(B, Cin, H, W, generator=g(7))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(8)) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g(9))
    with forced(variant):
        y = ctx.op_conv(x, w, b, stride=stride, pad=1, up=up)
    xr = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    check(f"dma2[{variant}]_conv3x3_{Cin}_{Cout}_{H}x{W}_s{stride}_up{int(up)}", y,
          F.conv2d(xr, w, b, stride=stride, padding=1), TOL)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("M,K,N,bias", [(1560, 320, 320, True), (390, 640, 640, False), (130, 2560, 640, True),
                                         (257, 64, 77, False), (16, 1280, 6080, True)])
def test_linear(ctx, variant, M, K, N, bias):
    a = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3)) if bias else None
    with forced(variant):
        y = ctx.op_linear(a, w, b)
    check(f"dma2[{variant}]_linear_{M}x{K}x{N}", y, F.linear(a, w, b), TOL)


@pytest.mark.parametrize("variant", VARIANTS)
def test_linear_geglu(ctx, variant):
    a = torch.randn(1560, 320, generator=g(4))
    w = torch.randn(2560, 320, generator=g(5)) / math.sqrt(320)
    b = torch.randn(2560, generator=g(6)) * 0.1
    val, gate = F.linear(a, w, b).chunk(2, dim=-1)
    with forced(variant):
        y = ctx.op_linear(a, w, b, geglu=True)
    check(f"dma2[{variant}]_geglu", y, val * F.gelu(gate), TOL)


def test_identity_asymmetric(ctx):
    """A = I against an asymmetric B: catches a transposed or mis-placed output block (fragment map, slab layout)."""
    K = N = 256
    a = torch.eye(K)
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 17.0 + torch.arange(N)[:, None] * 0.5
    for variant in VARIANTS:
        with forced(variant):
            y = ctx.op_linear(a, w)
        check(f"dma2[{variant}]_identity", y, w.t().contiguous(), TOL)


@pytest.mark.parametrize("variant", ["0,4,1,1"])
def test_one_slice_is_bit_identical_to_the_other_engines(ctx, variant):
    """Same products in the same order per accumulator: without a K split this engine, the 64x64 LDS-DMA engine and the
    register-staged engine agree bit for bit."""
    x = torch.randn(2, 320, 10, 78, generator=g(11))
    w = torch.randn(320, 320, 3, 3, generator=g(12)) / math.sqrt(2880)
    b = torch.randn(320, generator=g(13))
    with forced("off"):
        y_dma = ctx.op_conv(x, w, b, pad=1).cpu()
    with forced("off", presplit=False):
        y_reg = ctx.op_conv(x, w, b, pad=1).cpu()
    with forced(variant):
        y = ctx.op_conv(x, w, b, pad=1).cpu()
    assert torch.equal(y, y_dma)
    assert torch.equal(y, y_reg)


@pytest.mark.parametrize("variant", ["0,4,1,2", "0,4,1,3", None])
def test_split_k_is_deterministic_and_batch_invariant(ctx, variant):
    """Slices are added in slice order by a separate kernel and their number depends on the layer only: repeated runs
    are bit-identical, and a sample's rows do not change with the batch they are computed in (None = default policy)."""
    x = torch.randn(6, 640, 5, 39, generator=g(21))
    w = torch.randn(640, 640, 3, 3, generator=g(22)) / math.sqrt(5760)
    b = torch.randn(640, generator=g(23))
    env = forced(variant) if variant else forced("", presplit=True)
    with env:
        y6 = ctx.op_conv(x, w, b, pad=1).cpu()
        y6b = ctx.op_conv(x, w, b, pad=1).cpu()
        y1 = ctx.op_conv(x[4:5], w, b, pad=1).cpu()
    assert torch.equal(y6, y6b)
    assert torch.equal(y6[4:5], y1)
    check(f"dma2[{variant}]_conv_640_b6", y6, F.conv2d(x, w, b, padding=1), TOL)


@pytest.mark.parametrize("variant", ["0,4,1,1", "0,4,1,2"])
def test_persistent_workgroups_match_one_workgroup_per_item(ctx, variant):
    """A grid smaller than the number of (slice, tile) items: every workgroup runs several items as one stream of K
    chunks (the next item's copies are issued before this item's epilogue).  Batch 16 of the benchmark's 10x78 level has
    975 64x64 tiles -- several per workgroup; a sample's rows must be bit-identical to a launch small enough for one item per
    workgroup."""
    x = torch.randn(16, 320, 10, 78, generator=g(41))
    w = torch.randn(320, 320, 3, 3, generator=g(42)) / math.sqrt(2880)
    b = torch.randn(320, generator=g(43))
    with forced(variant):
        y = ctx.op_conv(x, w, b, pad=1).cpu()
        y1 = ctx.op_conv(x[:2], w, b, pad=1).cpu()      # 2 samples: fewer items than workgroup slots, one item per workgroup
    assert torch.equal(y[:2], y1)
    check(f"dma2[{variant}]_persistent_conv_b16", y, F.conv2d(x, w, b, padding=1), TOL)
    a = torch.randn(12480, 320, generator=g(44))
    wl = torch.randn(960, 320, generator=g(45)) / math.sqrt(320)
    with forced(variant):
        z = ctx.op_linear(a, wl).cpu()
    check(f"dma2[{variant}]_persistent_linear_12480x320x960", z, F.linear(a, wl), TOL)


def test_a_stale_override_is_refused_when_the_tuning_is_parsed(ctx):
    """MAA_DMA2 in round 2's format ("2,2,0,1" = tile, stages, pipelining, slices of an instantiation that no longer exists)
    must fail where the knobs are parsed -- context creation / reload_tuning -- with a message naming the variable, not as
    a check inside a forward pass; the context then runs on the default policy."""
    x = torch.randn(2, 320, 10, 78, generator=g(51))
    w = torch.randn(320, 320, 3, 3, generator=g(52)) / math.sqrt(2880)
    y0 = ctx.op_conv(x, w, None, pad=1).cpu()
    os.environ["MAA_DMA2"] = "2,2,0,1"
    try:
        with pytest.raises(RuntimeError, match="MAA_DMA2"):
            ctx.reload_tuning()
        y1 = ctx.op_conv(x, w, None, pad=1).cpu()
    finally:
        os.environ.pop("MAA_DMA2", None)
        ctx.reload_tuning()
    assert torch.equal(y0, y1)
