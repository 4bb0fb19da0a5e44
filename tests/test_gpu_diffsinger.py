"""DiffSinger's denoiser (DiffNet) and the PLMS loop on the device (csrc/diffnet.cpp, diffsinger.hip).

Golden: tests/golden/diffsinger_ds1000.npz, produced by the reference's DiffNet (NeuralSeq/modules/diff/net.py:84-130)
and GaussianDiffusion.p_sample_plms (shallow_diffusion_tts.py:166-201) over K_step = 60, pndm_speedup = 10: the
2-evaluation first step, then the 2nd, 3rd and 4th-order multistep formulas (6 steps).
"""
import numpy as np
import pytest
import torch

from audiogpt_amd import config as C
from audiogpt_amd import weights as WT
from tests.util import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("bf16x3", 2e-4)])
def test_diffnet_and_plms_match_reference(golden, precision, tol):
    from audiogpt_amd.backend import Context, DiffNet
    from oracle import diffsinger as D
    g = golden("diffsinger_ds1000")
    cfg = C.DIFFSINGER_DS1000
    ctx = Context("cuda:0", precision=precision)
    net = DiffNet(ctx, cfg, WT.make_diffnet_state_dict(cfg, seed=7))
    cond, x_T = torch.from_numpy(g["cond"]), torch.from_numpy(g["x_T"])
    eps0 = net(x_T, torch.from_numpy(g["t0"]), cond)
    check(f"{precision}_diffnet_eps0_vs_reference", eps0, g["eps0"], tol)
    ac = g["alphas_cumprod"]
    K, iv = int(g["K_step"]), cfg["pndm_speedup"]
    x0 = net.plms_sample(x_T, cond, ac, K, iv, use_graph=True)
    check(f"{precision}_plms_x0_vs_reference", x0, g["x0"], 5 * tol)
    x0_eager = net.plms_sample(x_T, cond, ac, K, iv, use_graph=False)
    assert torch.equal(x0, x0_eager), "graph replay differs from the eager loop"
    # every intermediate of the reference loop, driving the same device denoiser step by step from Python
    trace = []
    D.plms_sample(lambda x, t, c: net(x, t, c).cpu(), torch.from_numpy(ac), x_T, cond, K, iv, trace=trace)
    assert len(trace) == g["x_inter"].shape[0]
    for i, x in enumerate(trace):
        check(f"{precision}_plms_step{i}_vs_reference", x, g["x_inter"][i], 5 * tol)
    # batch rows are independent, and the conditioning cache follows the conditioning
    x2 = torch.cat([x_T, x_T.flip(-1)])
    c2 = torch.cat([cond, cond.flip(-1)])
    y2 = net.plms_sample(x2, c2, ac, K, iv)
    assert torch.equal(y2[:1], x0)
    from audiogpt_amd._lib import MaaError
    with pytest.raises(MaaError):
        net(x_T, torch.from_numpy(g["t0"]), cond[:, :100])
    net.close()
    ctx.close()


def test_gaussian_diffusion_infer_matches_oracle():
    """The reference surface (audiogpt_amd/diffsinger.GaussianDiffusion.infer = the infer branch of forward after fs2):
    norm_spec -> q_sample at K_step - 1 with the given noise -> PLMS loop (K_step 100, speed-up 10) -> denorm_spec."""
    from audiogpt_amd.diffsinger import GaussianDiffusion
    from oracle import diffsinger as D
    cfg = dict(C.DIFFSINGER_DS1000, K_step=100)
    gd = GaussianDiffusion(cfg, device="cuda:0", precision="f32")
    gen = torch.Generator().manual_seed(3)
    B, T = 2, 120
    fs2 = torch.rand(B, T, 80, generator=gen) * 6.0 - 5.0
    cond = torch.randn(B, 256, T, generator=gen)
    noise = torch.randn(B, 1, 80, T, generator=gen)
    mel = gd.infer(fs2, cond.cuda(), noise=noise.cuda()).cpu()
    assert mel.shape == (B, T, 80)
    sd = WT.make_diffnet_state_dict(cfg, seed=7)
    ac = D.alphas_cumprod(cfg["timesteps"], cfg["max_beta"])
    smin, smax = torch.full((1, 1, 80), -6.0), torch.full((1, 1, 80), 1.5)
    x0 = ((fs2 - smin) / (smax - smin) * 2 - 1).transpose(1, 2)[:, None]
    t = cfg["K_step"] - 1
    x = ac[t].sqrt() * x0 + (1 - ac[t]).sqrt() * noise
    with torch.no_grad():
        xr = D.plms_sample(lambda x_, t_, c_: D.diffnet_forward(sd, cfg, x_, t_, c_), ac, x, cond, cfg["K_step"], cfg["pndm_speedup"])
    ref = D.denorm_spec(xr, smin, smax)
    check("f32_GaussianDiffusion.infer_vs_oracle", mel, ref, 5e-4)
