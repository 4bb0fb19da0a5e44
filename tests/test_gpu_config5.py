"""BASELINE configs[4] in its one-GPU form -- the `mixed` workload bench.py times -- against the reference at the benched
shape: 8 inpaint clips (VAE encode, concat-conditioned DDIM [8,9,10,106] without guidance, decode, compositing, BigVGAN over
848 frames) + 8 image-to-audio clips (one-token context, guidance 3, BigVGAN over 624 frames), 100 DDIM steps each.
tests/golden/mixed_config5_s100.npz holds one row of each tool computed by the reference's own DDIMSampler / UNetModel
(both files) / Encoder / Decoder / BigVGAN on bench.mixed_inputs (tests/golden/make_golden.py mixed).

Gates: mel-L1 <= 1e-4 on the [0,1] mel, waveform RMS <= 1e-4, latent rel-max recorded and gated at
1e-3, in the benchmark's precision (bf16x3) and in exact fp32.  And the sharding contract for both tools: a clip computed
alone is bit-identical to the same clip inside the batch of 8 (latent, mel, waveform).
"""
import pytest
import torch

from tests.util import record, rel_err

pytestmark = pytest.mark.gpu


def _gates(tag, z, spec, wav, gz, gspec, gwav):
    rz, _, _ = rel_err(z, gz)
    l1 = float((spec.double() - torch.from_numpy(gspec).double()).abs().mean())
    rms = float(((wav.double() - torch.from_numpy(gwav).double()) ** 2).mean().sqrt())
    record(tag, latent_rel_max=rz, mel_l1=l1, wav_rms=rms, tol=1e-4)
    assert rz <= 1e-3, (tag, rz)
    assert l1 <= 1e-4, f"{tag}: mel-L1 {l1:.3e} misses the 1e-4 gate at 100 steps"
    assert rms <= 1e-4, f"{tag}: waveform RMS {rms:.3e} misses the 1e-4 gate at 100 steps"


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_mixed_batch8_100_steps_matches_reference(golden, precision):
    import bench
    from audiogpt_amd import config as C
    from audiogpt_amd.pipeline import MakeAnAudio
    g = golden("mixed_config5_s100")
    S, n = int(g["S"]), int(g["n_batch"])
    assert S == bench.DDIM_STEPS and n == bench.PROMPTS_PER_GPU
    dev = "cuda:0"
    mel, mask, emb, uc, noise, xT_inp, xT_i2a = (t.to(dev) for t in bench.mixed_inputs(n))

    # image-to-audio, batch of 8 with guidance 3 (UNet batch 16)
    i2a = MakeAnAudio(dev, ldm=C.LDM_I2A, vocoder_cfg=C.BIGVGAN_16K, seeds=(4, 1, 3), precision=precision)
    wav, spec, z = (t.cpu() for t in i2a.generate(xT_i2a, emb, uc, 3.0, S))
    assert wav.shape == (n, 624 * 256) and spec.shape == (n, 80, 624)
    r = int(g["row_i2a"])
    _gates(f"{precision}_config5_i2a_batch8_s100", z[r:r + 1], spec[r:r + 1], wav[r:r + 1], g["i2a_z"], g["i2a_spec"], g["i2a_wav"])
    w1, s1, z1 = (t.cpu() for t in i2a.generate(xT_i2a[r:r + 1], emb[r:r + 1], uc[:1], 3.0, S))
    assert torch.equal(z1, z[r:r + 1]), "image-to-audio: the latent of a clip depends on the batch it was sampled in"
    assert torch.equal(s1, spec[r:r + 1]) and torch.equal(w1, wav[r:r + 1])
    i2a.close()

    # inpainting, batch of 8, concat conditioning, no guidance
    inp = MakeAnAudio(dev, ldm=C.LDM_INPAINT, vocoder_cfg=C.BIGVGAN_16K, seeds=(5, 1, 3), with_encoder=True, precision=precision)
    wav, comp, z = (t.cpu() for t in bench.mixed_inpaint(inp, mel, mask, noise, xT_inp, S))
    assert wav.shape == (n, 848 * 256) and comp.shape == (n, 80, 848)
    r = int(g["row_inp"])
    _gates(f"{precision}_config5_inpaint_batch8_s100", z[r:r + 1], comp[r:r + 1], wav[r:r + 1], g["inp_z"], g["inp_spec"], g["inp_wav"])
    w1, c1, z1 = (t.cpu() for t in bench.mixed_inpaint(inp, mel[r:r + 1], mask[r:r + 1], noise[r:r + 1], xT_inp[r:r + 1], S))
    assert torch.equal(z1, z[r:r + 1]), "inpaint: the latent of a clip depends on the batch it was sampled in"
    assert torch.equal(c1, comp[r:r + 1]) and torch.equal(w1, wav[r:r + 1])
    inp.close()
