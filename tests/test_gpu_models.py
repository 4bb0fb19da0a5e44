"""Model-level parity of the HIP path (through the C ABI) against the reference-generated golden
vectors (tests/golden, made by the reference's own modules) and against the CPU oracle.

Stated tolerances (fp32 path): UNet eps rel-max 1e-4; 10-step DDIM latent rel-max 1e-3; mel-L1 on the
[0,1] mel <= 1e-4; waveform RMS error <= 1e-4 of full scale."""
import numpy as np
import pytest
import torch

from audiogpt_amd import config as C
from audiogpt_amd import weights as WT
from tests.util import check, record, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from audiogpt_amd.backend import Context
    c = Context("cuda:0")
    yield c
    c.close()


@pytest.fixture(scope="module")
def unet_t2a(ctx):
    from audiogpt_amd.backend import UNet
    return UNet(ctx, C.UNET_T2A, WT.make_unet_state_dict(C.UNET_T2A, seed=0))


@pytest.fixture(scope="module")
def vae(ctx):
    from audiogpt_amd.backend import VAE
    return VAE(ctx, C.VAE_DDCONFIG, WT.make_vae_state_dict(C.VAE_DDCONFIG, seed=1))


def _ddim_tables(S, ldm):
    from oracle import ddim as O
    ac = O.alphas_cumprod(ldm["timesteps"], ldm["linear_start"], ldm["linear_end"])
    steps = O.ddim_timesteps(S, ldm["timesteps"])
    a, ap, _, _ = O.ddim_tables(ac, steps)
    return steps, a.numpy(), ap.numpy()


def test_unet_t2a_matches_reference(golden, unet_t2a):
    g = golden("unet_t2a")
    y = unet_t2a(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["context"]))
    check("unet_t2a_vs_reference", y, g["y"], 1e-4)


def test_unet_batch_invariance(golden, unet_t2a):
    """Sharding prompts over GPUs must not change results: sample i of a batch == the same sample alone."""
    g = golden("unet_t2a")
    x, t, c = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["context"])
    xb = torch.cat([x, x.flip(0), x]), torch.cat([t, t.flip(0), t]), torch.cat([c, c.flip(0), c])
    yb = unet_t2a(*xb).cpu()
    y1 = unet_t2a(x[1:2], t[1:2], c[1:2]).cpu()
    assert torch.equal(yb[1:2], y1) and torch.equal(yb[2:3], y1), "UNet output depends on batch composition"


@pytest.mark.parametrize("name,cfg,seed", [("unet_i2a", C.UNET_I2A, 4), ("unet_inpaint", C.UNET_INPAINT, 5)])
def test_unet_variants_match_reference(golden, ctx, name, cfg, seed):
    from audiogpt_amd.backend import UNet
    g = golden(name)
    u = UNet(ctx, cfg, WT.make_unet_state_dict(cfg, seed=seed))
    c = torch.from_numpy(g["context"]) if "context" in g else None
    y = u(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), c)
    check(name + "_vs_reference", y, g["y"], 1e-4)
    u.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_ddim_10_steps_matches_reference(golden, unet_t2a, use_graph):
    g = golden("ddim_t2a_s10")
    steps, a, ap = _ddim_tables(int(g["S"]), C.LDM_T2A)
    assert steps.tolist() == g["ddim_timesteps"].tolist()
    z = unet_t2a.ddim_sample(torch.from_numpy(g["x_T"]), steps, a, ap, cond=torch.from_numpy(g["c"]),
                             uncond=torch.from_numpy(g["uc"]), scale=float(g["scale"]), use_graph=use_graph)
    check(f"ddim_s10_graph{int(use_graph)}_vs_reference", z, g["z"], 1e-3)


def test_ddim_graph_replay_is_bit_identical(golden, unet_t2a):
    g = golden("ddim_t2a_s10")
    steps, a, ap = _ddim_tables(6, C.LDM_T2A)
    args = dict(cond=torch.from_numpy(g["c"]), uncond=torch.from_numpy(g["uc"]), scale=1.5)
    z0 = unet_t2a.ddim_sample(torch.from_numpy(g["x_T"]), steps, a, ap, use_graph=False, **args).cpu()
    z1 = unet_t2a.ddim_sample(torch.from_numpy(g["x_T"]), steps, a, ap, use_graph=True, **args).cpu()
    assert torch.equal(z0, z1)


def test_ddim_step_graph_is_kept_across_calls_and_invalidated_by_its_inputs(golden, unet_t2a):
    """The captured step outlives sample(): a second call with other latents / conditioning VALUES (in fresh buffers) replays
    it and equals the eager loop; another guidance scale, batch or step count must not reuse it."""
    g = golden("ddim_t2a_s10")
    steps, a, ap = _ddim_tables(5, C.LDM_T2A)
    c, uc, x = torch.from_numpy(g["c"]), torch.from_numpy(g["uc"]), torch.from_numpy(g["x_T"])
    gen = torch.Generator().manual_seed(3)
    cases = [dict(x=x, c=c, uc=uc, scale=1.5, st=(steps, a, ap)),
             dict(x=torch.randn(x.shape, generator=gen), c=torch.randn(c.shape, generator=gen), uc=uc, scale=1.5, st=(steps, a, ap)),
             dict(x=x, c=c, uc=uc, scale=2.5, st=(steps, a, ap)),                                  # other guidance scale
             dict(x=x[:1], c=c[:1], uc=uc[:1], scale=2.5, st=(steps, a, ap)),                      # other batch
             dict(x=x, c=c, uc=uc, scale=1.5, st=_ddim_tables(7, C.LDM_T2A)),                      # other step count
             dict(x=x, c=c, uc=uc, scale=1.5, st=(steps, a, ap))]
    for i, k in enumerate(cases):
        kw = dict(cond=k["c"], uncond=k["uc"], scale=k["scale"])
        zg = unet_t2a.ddim_sample(k["x"].clone(), *k["st"], use_graph=True, **kw).cpu()
        ze = unet_t2a.ddim_sample(k["x"].clone(), *k["st"], use_graph=False, **kw).cpu()
        assert torch.equal(zg, ze), "case %d: graph replay differs from the eager loop" % i


def test_vae_decode_and_encode_match_reference(golden, vae):
    g = golden("vae")
    mel = vae.decode(torch.from_numpy(g["z"]), 1.0)
    check("vae_decode_vs_reference", mel, g["mel"], 2e-4)
    m01 = torch.clamp((mel.cpu() + 1) / 2, 0, 1)
    r01 = torch.clamp((torch.from_numpy(g["mel"]) + 1) / 2, 0, 1)
    l1 = float((m01 - r01).abs().mean())
    record("vae_mel_l1", mel_l1=l1, tol=1e-4)
    assert l1 <= 1e-4
    mom = vae.encode_moments(torch.from_numpy(g["mel_in"]))
    check("vae_encode_vs_reference", mom, g["moments"], 2e-4)
    # the tools' clamp((x + 1) / 2, 0, 1) folded into the decoder's last pass: the same values bit for bit, [B, 80, T]
    spec = vae.decode_spec(torch.from_numpy(g["z"]), 1.0)
    assert torch.equal(spec, torch.clamp((mel + 1.0) / 2.0, min=0.0, max=1.0)[:, 0])


@pytest.mark.parametrize("name,cfg", [("hifigan_16k_t2a", C.HIFIGAN_16K), ("hifigan_ns512", C.HIFIGAN_NS_512),
                                      ("hifigan_ns128", C.HIFIGAN_NS_128)])
def test_hifigan_matches_reference(golden, ctx, name, cfg):
    from audiogpt_amd.backend import Vocoder
    g = golden(name)
    v = Vocoder(ctx, cfg, WT.make_vocoder_state_dict(cfg, seed=2))
    wav = v(torch.from_numpy(g["mel"])).cpu()
    ref = torch.from_numpy(g["wav"])
    rms = float(((wav - ref) ** 2).mean().sqrt())
    record(name + "_wav_rms", wav_rms=rms, tol=1e-4)
    check(name + "_vs_reference", wav, ref, 2e-4)
    assert rms <= 1e-4
    v.close()


def test_bigvgan_matches_reference(golden, ctx):
    from audiogpt_amd.backend import Vocoder
    g = golden("bigvgan_16k")
    v = Vocoder(ctx, C.BIGVGAN_16K, WT.make_vocoder_state_dict(C.BIGVGAN_16K, seed=3))
    wav = v(torch.from_numpy(g["mel"])).cpu()
    ref = torch.from_numpy(g["wav"])
    rms = float(((wav - ref) ** 2).mean().sqrt())
    record("bigvgan_wav_rms", wav_rms=rms, tol=1e-4)
    check("bigvgan_vs_reference", wav, ref, 5e-4)
    assert rms <= 1e-4
    v.close()


@pytest.mark.parametrize("name,cfg,seed,tol", [("hifigan_rb2", C.HIFIGAN_RB2, 12, 2e-4), ("bigvgan_rb2", C.BIGVGAN_RB2, 13, 5e-4)])
def test_resblock2_generators_match_reference(golden, ctx, name, cfg, seed, tol):
    """`resblock: "2"` (VERDICT r5 missing #5): ResBlock2 (hifigan.py:70-91, modules.py:62-83) and AMPBlock2 with plain
    `snake` (bigvgan/models.py:90-132) against the reference generators' own outputs."""
    from audiogpt_amd.backend import Vocoder
    g = golden(name)
    v = Vocoder(ctx, cfg, WT.make_vocoder_state_dict(cfg, seed=seed))
    wav = v(torch.from_numpy(g["mel"])).cpu()
    ref = torch.from_numpy(g["wav"])
    rms = float(((wav - ref) ** 2).mean().sqrt())
    record(name + "_wav_rms", wav_rms=rms, tol=1e-4)
    check(name + "_vs_reference", wav, ref, tol)
    assert rms <= 1e-4
    v.close()


def test_t2a_plumbing_config1_end_to_end(golden, ctx, unet_t2a, vae):
    """BASELINE config 1: 1 prompt, 10 DDIM steps, CFG 1.5 -> VAE -> clamp -> HiFi-GAN, vs the reference chain."""
    from audiogpt_amd.backend import Vocoder
    g = golden("ddim_t2a_s10")
    steps, a, ap = _ddim_tables(10, C.LDM_T2A)
    z = unet_t2a.ddim_sample(torch.from_numpy(g["x_T"]), steps, a, ap, cond=torch.from_numpy(g["c"]),
                             uncond=torch.from_numpy(g["uc"]), scale=1.5)
    mel = vae.decode(z, 1.0)
    spec = torch.clamp((mel + 1.0) / 2.0, 0.0, 1.0)[:, 0]
    ref_spec = torch.from_numpy(golden("hifigan_16k_t2a")["mel"])
    l1 = float((spec.cpu() - ref_spec).abs().mean())
    record("t2a_config1_mel_l1", mel_l1=l1, tol=1e-4)
    assert l1 <= 1e-4, f"mel-L1 {l1:.3e}"
    v = Vocoder(ctx, C.HIFIGAN_16K, WT.make_vocoder_state_dict(C.HIFIGAN_16K, seed=2))
    wav = v(spec).cpu()
    ref = torch.from_numpy(golden("hifigan_16k_t2a")["wav"])
    rms = float(((wav - ref) ** 2).mean().sqrt())
    record("t2a_config1_wav_rms", wav_rms=rms, tol=1e-4)
    assert rms <= 1e-4, f"waveform RMS error {rms:.3e}"
    v.close()


def test_vocoder_batch_invariance_and_ragged_lengths(ctx):
    """Edge cases: T not a multiple of any tile, batch rows independent."""
    from audiogpt_amd.backend import Vocoder
    from oracle import vocoder as O
    cfg = C.HIFIGAN_NS_128
    sd = WT.make_vocoder_state_dict(cfg, seed=2)
    v = Vocoder(ctx, cfg, sd)
    gen = torch.Generator().manual_seed(11)
    for T in (1, 7, 33):
        mel = torch.randn(3, 80, T, generator=gen)
        wav = v(mel).cpu()
        with torch.no_grad():
            ref = O.hifigan_forward(O.fold_weight_norm(sd), cfg, mel)
        check(f"hifigan_T{T}", wav, ref, 2e-4)
        one = v(mel[2:3]).cpu()
        assert torch.equal(one, wav[2:3])
    v.close()


def _ddim_variant(ctx, golden, name, cfg, ldm, seed, tol, tag=""):
    """The device DDIM loop on the I2A (1-token context, scale 3) and inpaint (conditioning_key 'concat', no guidance)
    call patterns against the reference sampler's output (tests/golden/make_golden.py ddim_variant_case)."""
    from audiogpt_amd.backend import UNet
    g = golden(name)
    u = UNet(ctx, cfg, WT.make_unet_state_dict(cfg, seed=seed))
    steps, a, ap = _ddim_tables(int(g["S"]), ldm)
    assert steps.tolist() == g["ddim_timesteps"].tolist()
    if ldm["conditioning_key"] == "concat":
        kw = dict(concat=torch.from_numpy(g["c"]))
    else:
        kw = dict(cond=torch.from_numpy(g["c"]), uncond=torch.from_numpy(g["uc"]), scale=float(g["scale"]))
    z = u.ddim_sample(torch.from_numpy(g["x_T"]), steps, a, ap, **kw)
    z_eager = u.ddim_sample(torch.from_numpy(g["x_T"]), steps, a, ap, use_graph=False, **kw)
    u.close()
    assert torch.equal(z, z_eager), "graph replay differs from eager"
    check(tag + name + "_vs_reference", z, g["z"], tol)


@pytest.mark.parametrize("name,cfg,ldm,seed", [("ddim_i2a_s4", C.UNET_I2A, C.LDM_I2A, 4),
                                               ("ddim_inpaint_s4", C.UNET_INPAINT, C.LDM_INPAINT, 5)])
def test_ddim_variants_match_reference(golden, ctx, name, cfg, ldm, seed):
    _ddim_variant(ctx, golden, name, cfg, ldm, seed, 1e-4)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_cfg_halves_as_two_lanes_are_bit_identical_to_one_stream(golden, precision):
    """csrc/ddim.cpp runs the unconditional and the conditional half of a classifier-free-guidance step (ddim.py:177-199: one
    model call on cat([x] * 2)) as two lanes -- two branches of the captured step graph, the second on its own stream and
    workspace.  Every kernel is batch-invariant, so the latents must equal the one-stream form's bit for bit: T2A (77-token
    context, scale 1.5) and the image-to-audio UNet (1-token context added to the time embedding, scale 3), graph and eager, a
    second call on the kept graph, and a batch of one."""
    from audiogpt_amd.backend import Context, UNet
    c = Context("cuda:0", precision=precision)
    try:
        for name, cfg, ldm, seed in (("ddim_t2a_s10", C.UNET_T2A, C.LDM_T2A, 0), ("ddim_i2a_s4", C.UNET_I2A, C.LDM_I2A, 4)):
            g = golden(name)
            u = UNet(c, cfg, WT.make_unet_state_dict(cfg, seed=seed))
            steps, a, ap = _ddim_tables(4, ldm)
            x = torch.from_numpy(g["x_T"])
            kw = dict(cond=torch.from_numpy(g["c"]), uncond=torch.from_numpy(g["uc"]), scale=float(g["scale"]))
            out = {}
            for lanes in (True, False):
                c.set_cfg_split(lanes)
                out[lanes, "eager"] = u.ddim_sample(x, steps, a, ap, use_graph=False, **kw).cpu()
                out[lanes, "graph"] = u.ddim_sample(x, steps, a, ap, use_graph=True, **kw).cpu()
                out[lanes, "again"] = u.ddim_sample(x, steps, a, ap, use_graph=True, **kw).cpu()
                one = u.ddim_sample(x[:1], steps, a, ap, use_graph=True, cond=kw["cond"][:1], uncond=kw["uncond"][:1], scale=kw["scale"]).cpu()
                assert torch.equal(one, out[lanes, "graph"][:1]), (name, lanes)
            ref = out[False, "eager"]
            assert float(ref.abs().max()) > 0 and bool(torch.isfinite(ref).all())
            for k, v in out.items():
                assert torch.equal(v, ref), (name, k)
            u.close()
        c.set_cfg_split(None)
    finally:
        c.close()




@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_guided_step_shares_the_layers_before_the_first_cross_attention(golden, precision):
    """Round 6: a guided step evaluates the model on cat([x] * 2) (ddim.py:177-199); the halves differ only in the context, so
    conv_in, the first ResBlock and the first transformer's norm / proj_in / self-attention / to_q are computed ONCE on the
    one-stream form (unet.cpp `dup`) and duplicated where the halves part.  MAA_CFG_SHARED=0 evaluates both halves in full (what
    rounds 1-5 did): the latent must be the same bit for bit -- plain, with mask + eta > 0 + intermediates, graph and eager, a
    batch of one -- and the profiler must see fewer attention / convolution FLOPs with the shared prefix."""
    import os
    from audiogpt_amd.backend import Context, UNet, reload_tuning
    g, gm = golden("ddim_t2a_s10"), golden("ddim_t2a_mask_eta_s6")
    c = Context("cuda:0", precision=precision)
    c.set_cfg_split(False)
    u = UNet(c, C.UNET_T2A, WT.make_unet_state_dict(C.UNET_T2A, seed=0))
    steps, a, ap = _ddim_tables(4, C.LDM_T2A)
    t = lambda k: torch.from_numpy(gm[k])
    from oracle import ddim as O
    ac = O.alphas_cumprod(C.LDM_T2A["timesteps"], C.LDM_T2A["linear_start"], C.LDM_T2A["linear_end"])
    sig = O.ddim_tables(ac, steps, 0.5)[2].numpy()
    sq_ac, sq_1mac = (v.numpy() for v in O.q_sample_tables(ac, steps))
    full = dict(cond=t("c"), uncond=t("uc"), scale=1.5, mask=t("mask"), x0=t("x0"), noise_q=t("noise_q")[:len(steps)],
                sqrt_ac=sq_ac, sqrt_1mac=sq_1mac, sigmas=sig, noise_p=t("noise_p")[:len(steps)], temperature=0.9, log_every_t=1)
    plain = dict(cond=torch.from_numpy(g["c"]), uncond=torch.from_numpy(g["uc"]), scale=float(g["scale"]))
    out, flops = {}, {}
    try:
        for shared in ("1", "0"):
            os.environ["MAA_CFG_SHARED"] = shared
            reload_tuning()
            x = torch.from_numpy(g["x_T"])
            for lanes in (False, True):      # one stream: one half computed and duplicated; two lanes: the second starts from the first's prefix
                c.set_cfg_split(lanes)
                out[shared, lanes, "eager"] = u.ddim_sample(x, steps, a, ap, use_graph=False, **plain).cpu()
                out[shared, lanes, "graph"] = u.ddim_sample(x, steps, a, ap, use_graph=True, **plain).cpu()
                out[shared, lanes, "again"] = u.ddim_sample(x, steps, a, ap, use_graph=True, **plain).cpu()
                out[shared, lanes, "one"] = u.ddim_sample(x[:1], steps, a, ap, use_graph=True, cond=plain["cond"][:1], uncond=plain["uncond"][:1],
                                                          scale=plain["scale"]).cpu()
                z, xi, p0 = u.ddim_sample(t("x_T"), steps, a, ap, use_graph=True, **full)
                out[shared, lanes, "full"] = torch.cat([z.cpu().flatten(), xi.cpu().flatten(), p0.cpu().flatten()])
                c.prof_begin()
                u.ddim_sample(x, steps[:1], a[:1], ap[:1], use_graph=False, **plain)
                rows = c.prof_end()
                flops[shared, lanes] = sum(v["flops"] for v in rows.values())
    finally:
        os.environ.pop("MAA_CFG_SHARED", None)
        reload_tuning()
        c.set_cfg_split(None)
    ref = {k: out["0", False, k] for k in ("eager", "graph", "again", "one", "full")}
    assert float(ref["eager"].abs().max()) > 0 and all(bool(torch.isfinite(v).all()) for v in ref.values())
    assert torch.equal(ref["eager"], ref["graph"]) and torch.equal(ref["one"], ref["graph"][:1])
    for (shared, lanes, k), v in out.items():
        assert torch.equal(v, ref[k]), (shared, lanes, k)
    # conv_in + two 320 -> 320 convolutions + four 320-wide linears + one 780-token self-attention on half the batch
    for lanes in (False, True):
        assert 0.93 * flops["0", lanes] < flops["1", lanes] < 0.985 * flops["0", lanes], flops
    u.close()
    c.close()


def test_serving_arrangement_hint_changes_tiles_and_lanes_but_not_results(golden):
    """maa_ctx_set_concurrency (round 6): told that three contexts are kept in flight, a context takes the short-K contractions'
    tile by least total workgroup time (128 x 64 / 128 x 128 where a launch alone takes 64 x 64) and runs a guided DDIM step on one
    stream; told that it owns the GPU, 64 x 64 tiles and two CFG lanes.  No K split is involved: the UNet's output and a guided
    trajectory are the same bit for bit, a batch of 16 (the benchmark's) and the golden's batch alike."""
    from audiogpt_amd.backend import Context, UNet
    g = golden("unet_t2a")
    x, t, c = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["context"])
    reps = -(-16 // x.shape[0])
    x16, t16, c16 = (torch.cat([v] * reps)[:16] for v in (x, t, c))
    ctx = Context("cuda:0", precision="bf16x3")
    u = UNet(ctx, C.UNET_T2A, WT.make_unet_state_dict(C.UNET_T2A, seed=0))
    gd = golden("ddim_t2a_s10")
    steps, a, ap = _ddim_tables(4, C.LDM_T2A)
    kw = dict(cond=torch.from_numpy(gd["c"]), uncond=torch.from_numpy(gd["uc"]), scale=float(gd["scale"]))
    out, fams, traj = {}, {}, {}
    try:
        for n in (1, 3):
            ctx.set_concurrency(n)
            ctx.prof_begin()
            out[n] = u(x16, t16, c16).cpu()
            fams[n] = {k: v["launches"] for k, v in ctx.prof_end().items() if k.startswith("igemm_dma_bf16x3")}
            traj[n] = u.ddim_sample(torch.from_numpy(gd["x_T"]), steps, a, ap, use_graph=True, **kw).cpu()
        ctx.set_concurrency(None)
    finally:
        u.close()
        ctx.close()
    assert fams[1].get("igemm_dma_bf16x3<64x64>", 0) > fams[3].get("igemm_dma_bf16x3<64x64>", 0), fams
    assert fams[3].get("igemm_dma_bf16x3<128x64>", 0) + fams[3].get("igemm_dma_bf16x3<128x128>", 0) > \
        fams[1].get("igemm_dma_bf16x3<128x64>", 0) + fams[1].get("igemm_dma_bf16x3<128x128>", 0), fams
    assert torch.equal(out[1], out[3]) and torch.equal(traj[1], traj[3])
    check("unet_t2a_kept_full_tiles_vs_reference", out[3][:g["y"].shape[0]], g["y"], 2e-4)


def test_conv_item_order_does_not_change_results(golden):
    """The (K slice, tile) work items of a split-K convolution launch are walked slice-major (default since round 6: an XCD's
    contiguous eighth of the items belongs to one K slice, so its L2 streams 1 / S of the packed weights) or tile-major (round
    5, MAA_PP_TILE_MAJOR=1).  Scheduling only: the UNet's output is the same bit for bit."""
    import os
    from audiogpt_amd.backend import Context, UNet, reload_tuning
    g = golden("unet_t2a")
    x, t, c = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["context"])
    x, t, c = torch.cat([x, x.flip(0), x]), torch.cat([t, t.flip(0), t]), torch.cat([c, c.flip(0), c])
    ctx = Context("cuda:0", precision="bf16x3")
    u = UNet(ctx, C.UNET_T2A, WT.make_unet_state_dict(C.UNET_T2A, seed=0))
    out = {}
    try:
        for mode in ("0", "1"):
            os.environ["MAA_PP_TILE_MAJOR"] = mode
            reload_tuning()
            out[mode] = u(x, t, c).cpu()
    finally:
        os.environ.pop("MAA_PP_TILE_MAJOR", None)
        reload_tuning()
    assert torch.equal(out["0"], out["1"])
    check("unet_t2a_slice_major_vs_reference", out["0"][:g["y"].shape[0]], g["y"], 2e-4)
    u.close()
    ctx.close()
