"""The row-chain engine (csrc/rowchain.hip): linear -> LayerNorm -> linear per 64-row block, against plain PyTorch fp32.

Reference ops (text_to_audio/Make_An_Audio/ldm/modules/attention.py): SpatialTransformer.proj_in -> BasicTransformerBlock.norm1
-> attn1.to_q/k/v (:250-261, 203, 212), attn1.to_out(+x) -> norm2 -> attn2.to_q (:212-213), attn2.to_out(+x) -> norm3 (:213-214),
ff.net.2(+x) -> proj_out(+x_in) (:214, 259-261).  Tolerances are the operator-level ones of test_gpu_precision.py."""
import math
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from tests.util import check

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": 2e-4, "bf16": 5e-2}


@pytest.fixture(scope="module", params=["bf16x3", "bf16"])
def ctx(request):
    from audiogpt_amd.backend import Context
    c = Context("cuda:0", precision=request.param)
    yield c
    c.close()


def g(seed):
    return torch.Generator().manual_seed(seed)


def _case(M, K1, N, N2, seed):
    a = torch.randn(M, K1, generator=g(seed)) * 1.5 + 0.3
    w1 = torch.randn(N, K1, generator=g(seed + 1)) / math.sqrt(K1)
    b1 = torch.randn(N, generator=g(seed + 2)) * 0.2
    res1 = torch.randn(M, N, generator=g(seed + 3))
    gamma = 1.0 + 0.3 * torch.randn(N, generator=g(seed + 4))
    beta = 0.2 * torch.randn(N, generator=g(seed + 5))
    w2 = torch.randn(N2, N, generator=g(seed + 6)) / math.sqrt(N) if N2 else None
    b2 = torch.randn(N2, generator=g(seed + 7)) * 0.2 if N2 else None
    res2 = torch.randn(M, N, generator=g(seed + 8))
    return a, w1, b1, res1, gamma, beta, w2, b2, res2


# (M, K1, N, N2): the four chains of a 10 x 78 transformer block at batch 2 (M = 1560) and ragged row counts
@pytest.mark.parametrize("M,K1,N,N2", [(1560, 320, 320, 960), (1560, 320, 320, 320), (1000, 1280, 320, 320), (63, 320, 320, 320),
                                       (130, 256, 256, 768), (65, 1024, 256, 256), (1, 320, 320, 960)])
def test_linear_layernorm_linear(ctx, M, K1, N, N2):
    a, w1, b1, res1, gamma, beta, w2, b2, res2 = _case(M, K1, N, N2, 100 + M)
    y, t, z = ctx.op_rowchain(a, w1, b1, res1=res1, ln=(gamma, beta), w2=w2, b2=None if N2 != N else b2, want_t=True)
    y_ref = F.linear(a, w1, b1) + res1
    t_ref = F.layer_norm(y_ref, (N,), gamma, beta, 1e-5)
    z_ref = F.linear(t_ref, w2, None if N2 != N else b2)
    tol = TOL[ctx.precision]
    check(f"{ctx.precision}_rowchain_y_{M}x{K1}x{N}", y, y_ref, tol)
    check(f"{ctx.precision}_rowchain_ln_{M}x{K1}x{N}", t, t_ref, 2 * tol)
    check(f"{ctx.precision}_rowchain_z_{M}x{K1}x{N}x{N2}", z, z_ref, 2 * tol)


def test_no_layernorm_two_residuals(ctx):
    """ff.net.2(+x) -> proj_out(+x_in): no normalisation between the stages, a residual on both."""
    M, K1, N = 1560, 1280, 320
    a, w1, b1, res1, _, _, w2, b2, res2 = _case(M, K1, N, N, 7)
    y, t, z = ctx.op_rowchain(a, w1, b1, res1=res1, w2=w2, b2=b2, res2=res2, want_y=False, want_t=True)
    assert y is None
    y_ref = F.linear(a, w1, b1) + res1
    tol = TOL[ctx.precision]
    check(f"{ctx.precision}_rowchain_noln_t", t, y_ref, tol)
    check(f"{ctx.precision}_rowchain_noln_z", z, F.linear(y_ref, w2, b2) + res2, 2 * tol)


def test_single_stage_layernorm_out(ctx):
    """attn2.to_out(+x) -> norm3: one contraction, the normalised rows leave as split32."""
    M, K1, N = 777, 320, 320
    a, w1, b1, res1, gamma, beta, _, _, _ = _case(M, K1, N, 0, 11)
    y, t, z = ctx.op_rowchain(a, w1, b1, res1=res1, ln=(gamma, beta), want_t=True)
    assert z is None
    y_ref = F.linear(a, w1, b1) + res1
    tol = TOL[ctx.precision]
    check(f"{ctx.precision}_rowchain_single_y", y, y_ref, tol)
    check(f"{ctx.precision}_rowchain_single_ln", t, F.layer_norm(y_ref, (N,), gamma, beta, 1e-5), 2 * tol)


def test_stage1_bit_identical_to_the_plain_gemm_and_rows_independent(ctx):
    """Stage 1 issues its products in the order of the other bf16 engines: y equals op_linear's result + the residual bit for bit;
    and a row's results do not depend on which rows share its launch (batch invariance)."""
    M, K1, N = 640, 320, 320
    a, w1, b1, res1, gamma, beta, w2, _, _ = _case(M, K1, N, 960, 21)
    y, t, z = ctx.op_rowchain(a, w1, b1, res1=res1, ln=(gamma, beta), w2=w2, want_t=True)
    plain = ctx.op_linear(a, w1, b1) + res1.to(y.device)
    assert torch.equal(y, plain)
    y2, t2, z2 = ctx.op_rowchain(a[37:300], w1, b1, res1=res1[37:300], ln=(gamma, beta), w2=w2, want_t=True)
    assert torch.equal(y2, y[37:300]) and torch.equal(t2, t[37:300]) and torch.equal(z2, z[37:300])
    y3, t3, z3 = ctx.op_rowchain(a, w1, b1, res1=res1, ln=(gamma, beta), w2=w2, want_t=True)
    assert torch.equal(z3, z) and torch.equal(t3, t)


def test_layernorm_of_rows_with_a_large_mean(ctx):
    """Two-pass statistics: rows whose mean dwarfs their spread keep their precision."""
    M, K1, N = 200, 320, 320
    a, w1, b1, res1, gamma, beta, w2, _, _ = _case(M, K1, N, 320, 31)
    res1 = res1 + 300.0
    y, t, z = ctx.op_rowchain(a, w1, b1, res1=res1, ln=(gamma, beta), want_t=True)
    y_ref = F.linear(a, w1, b1) + res1
    # the reference normalises the fp32 y the kernel produced (the 300 offset makes y's own rounding the dominant term)
    t_ref = F.layer_norm(y.cpu().double(), (N,), gamma.double(), beta.double(), 1e-5)
    check(f"{ctx.precision}_rowchain_bigmean_ln", t, t_ref, 2e-4 if ctx.precision == "bf16x3" else 5e-2)
    check(f"{ctx.precision}_rowchain_bigmean_y", y, y_ref, TOL[ctx.precision])


_AB = r"""
import sys, torch
sys.path.insert(0, %r)
from audiogpt_amd import config as C, weights as WT
from audiogpt_amd.backend import Context, UNet
ctx = Context("cuda:0", precision="bf16x3")
cfg = C.UNET_T2A
unet = UNet(ctx, cfg, WT.make_unet_state_dict(cfg, seed=0))
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 4, 10, 78, generator=g)
t = torch.tensor([981.0, 401.0])
c = torch.nn.functional.layer_norm(torch.randn(2, 77, 1024, generator=g), (1024,))
eps = unet(x, t, c)
torch.save(eps.cpu(), sys.argv[1])
"""


def test_unet_with_and_without_the_chains_agree(tmp_path):
    """The UNet's eps with the four chains per transformer block against the launch-per-layer path (MAA_ROWCHAIN=0): the
    contractions are bit-identical, only the LayerNorm's summation order differs -> agreement far inside the parity gate."""
    outs = []
    for flag in ("1", "0"):
        out = str(tmp_path / ("eps%s.pt" % flag))
        env = dict(os.environ, MAA_ROWCHAIN=flag)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run([sys.executable, "-c", _AB % root, out], check=True, env=env, timeout=600)
        outs.append(torch.load(out))
    from tests.util import record, rel_err
    r, mean, mx = rel_err(outs[0], outs[1])
    record("bf16x3_unet_rowchain_vs_unfused", rel_max=r, abs_mean=mean, abs_max=mx, tol=2e-5)
    assert r <= 2e-5, r
