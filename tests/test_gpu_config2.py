"""BASELINE configs[1] itself -- the workload bench.py times -- against the reference: 8 prompts, 100 DDIM steps with CFG
1.5, VAE decode, HiFi-GAN (hifi_0127).  tests/golden/t2a_config2_s100.npz holds rows 0 and 5 of that batch computed by
the reference's own DDIMSampler / UNetModel / Decoder / Generator (tests/golden/make_golden.py config2_case).

Gates: mel-L1 <= 1e-4 on the [0,1] mel and waveform RMS <= 1e-4, in the benchmark's precision
mode (bf16x3) and in exact fp32; the latent's rel-max error is recorded and gated at 1e-3.  And the sharding contract:
a prompt computed alone is bit-identical to the same prompt inside the batch of 8 (latent, mel and waveform).
"""
import numpy as np
import pytest
import torch

from tests.util import record, rel_err

pytestmark = pytest.mark.gpu


def _inputs():
    from bench import LATENT, PROMPTS_PER_GPU, synth_conditioning
    n = PROMPTS_PER_GPU
    x_T = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float()
    c = synth_conditioning(n, 1234)
    uc = synth_conditioning(1, 1235).expand(n, -1, -1).contiguous()
    return x_T, c, uc


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_config2_batch8_100_steps_matches_reference(golden, precision):
    from audiogpt_amd.pipeline import MakeAnAudio
    from bench import CFG_SCALE, DDIM_STEPS
    g = golden("t2a_config2_s100")
    assert int(g["S"]) == DDIM_STEPS and float(g["scale"]) == CFG_SCALE and int(g["n_batch"]) == 8
    rows = [int(r) for r in g["rows"]]
    pipe = MakeAnAudio("cuda:0", precision=precision)
    x_T, c, uc = _inputs()
    wav, spec, z = pipe.generate(x_T, c, uc, CFG_SCALE, DDIM_STEPS)
    wav, spec, z = wav.cpu(), spec.cpu(), z.cpu()
    assert wav.shape == (8, 624 * 256) and spec.shape == (8, 80, 624)
    rz, _, _ = rel_err(z[rows], g["z"])
    l1 = float((spec[rows].double() - torch.from_numpy(g["spec"]).double()).abs().mean())
    rms = float(((wav[rows].double() - torch.from_numpy(g["wav"]).double()) ** 2).mean().sqrt())
    record(f"{precision}_config2_batch8_s100", latent_rel_max=rz, mel_l1=l1, wav_rms=rms, tol=1e-4)
    assert rz <= 1e-3, rz
    assert l1 <= 1e-4, f"{precision} mel-L1 {l1:.3e} misses the 1e-4 gate at 100 steps"
    assert rms <= 1e-4, f"{precision} waveform RMS {rms:.3e} misses the 1e-4 gate at 100 steps"
    # prompt 5 alone == prompt 5 in the batch of 8, bit for bit (what prompt sharding over GPUs relies on)
    r = rows[-1]
    wav1, spec1, z1 = pipe.generate(x_T[r:r + 1], c[r:r + 1], uc[:1], CFG_SCALE, DDIM_STEPS)
    assert torch.equal(z1.cpu(), z[r:r + 1]), "latent of a prompt depends on the batch it was sampled in"
    assert torch.equal(spec1.cpu(), spec[r:r + 1]) and torch.equal(wav1.cpu(), wav[r:r + 1])
    pipe.close()


def test_bigvgan_624_frames_matches_reference(golden):
    """The T2A / I2A / inpaint tools vocode 624-frame (848 for inpaint) mels through BigVGAN (audio-chatgpt.py:145,
    179-181); the small golden is 48 frames."""
    from audiogpt_amd import config as C
    from audiogpt_amd import weights as WT
    from audiogpt_amd.backend import Context, Vocoder
    g = golden("bigvgan_16k_t624")
    for precision in ("bf16x3", "f32"):
        ctx = Context("cuda:0", precision=precision)
        v = Vocoder(ctx, C.BIGVGAN_16K, WT.make_vocoder_state_dict(C.BIGVGAN_16K, seed=3))
        wav = v(torch.from_numpy(g["mel"])).cpu()
        ref = torch.from_numpy(g["wav"])
        rms = float(((wav.double() - ref.double()) ** 2).mean().sqrt())
        r, _, _ = rel_err(wav, ref)
        record(f"{precision}_bigvgan_t624", wav_rms=rms, rel_max=r, tol=1e-4)
        assert rms <= 1e-4 and r <= 5e-4, (precision, rms, r)
        v.close()
        ctx.close()
