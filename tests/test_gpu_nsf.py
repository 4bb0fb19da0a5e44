"""the NSF (f0-conditioned) branch of NeuralSeq's HiFi-GAN on the device (csrc/nsf.hip + the strided
noise_convs as implicit GEMMs), through `maa_vocoder_forward_f0`.

Reference: HifiGanGenerator(use_pitch_embed).forward(x, f0) (NeuralSeq/modules/hifigan/hifigan.py:144-169) with
SourceModuleHnNSF / SineGen (modules/parallel_wavegan/models/source.py:399-436, 526-535).  The golden
(tests/golden/hifigan_nsf_24k.npz) was produced by those modules; SineGen's two random draws are re-drawn here from the
golden's seed in the reference's order (oracle.nsf.draw_source_noise) and handed over the ABI.
Gates: waveform rel-max 2e-4, RMS 1e-4, in exact fp32 and bf16x3; longer clips and the source alone against the oracle.
"""
import numpy as np
import pytest
import torch

from audiogpt_amd import config as C
from audiogpt_amd import weights as WT
from tests.util import check, record

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_hifigan_nsf_matches_reference(golden, precision):
    from audiogpt_amd.backend import Context, Vocoder
    from oracle import nsf as N
    g = golden("hifigan_nsf_24k")
    cfg = C.HIFIGAN_NSF_24K
    hop = int(np.prod(cfg["upsample_rates"]))
    mel, f0 = torch.from_numpy(g["mel"]), torch.from_numpy(g["f0"])
    rand_ini, noise = N.draw_source_noise(int(g["noise_seed"]), mel.shape[0], mel.shape[2] * hop)
    ctx = Context("cuda:0", precision=precision)
    sd = WT.make_vocoder_state_dict(cfg, seed=6)
    v = Vocoder(ctx, cfg, sd)
    wav = v(mel, f0, rand_ini=rand_ini, noise=noise).cpu()
    ref = torch.from_numpy(g["wav"])
    rms = float(((wav.double() - ref.double()) ** 2).mean().sqrt())
    record(f"{precision}_hifigan_nsf_24k", wav_rms=rms, tol=1e-4)
    check(f"{precision}_hifigan_nsf_24k_vs_reference", wav, ref, 2e-4)
    assert rms <= 1e-4
    # batch rows are independent
    one = v(mel[1:2], f0[1:2], rand_ini=rand_ini[1:2], noise=noise[1:2]).cpu()
    assert torch.equal(one, wav[1:2])
    # without f0 the generator skips its source branch (HifiGanGenerator.forward(x, f0=None), hifigan.py:144-169)
    from oracle import vocoder as O_voc
    with torch.no_grad():
        plain = O_voc.hifigan_forward(O_voc.fold_weight_norm(sd), cfg, mel)
    check(f"{precision}_hifigan_nsf_24k_called_without_f0_vs_oracle", v.forward(mel).cpu(), plain, 2e-4)
    v.close()
    ctx.close()


def test_nsf_long_clip_with_unvoiced_gaps_matches_oracle():
    """600 frames (76 800 samples at hop 128): the phase accumulates over ~10^5 samples, f0 sweeps 80..700 Hz with
    unvoiced gaps (f0 = 0), harmonics wrap thousands of times."""
    from audiogpt_amd.backend import Context, Vocoder
    from oracle import nsf as N
    from oracle import vocoder as O_voc
    cfg = C.HIFIGAN_NSF_24K
    hop = int(np.prod(cfg["upsample_rates"]))
    B, T = 2, 600
    gen = torch.Generator().manual_seed(31)
    mel = torch.clamp(torch.randn(B, 80, T, generator=gen) * 1.5 - 2.25, -6.0, 1.5)
    t = torch.arange(T, dtype=torch.float32)
    f0 = torch.stack([80.0 + 620.0 * (0.5 + 0.5 * torch.sin(t * 0.021)), 220.0 + 100.0 * torch.cos(t * 0.05)])
    f0[0, 100:140] = 0.0
    f0[0, 400:401] = 0.0
    f0[1, :30] = 0.0
    f0[1, 550:] = 0.0
    rand_ini, noise = N.draw_source_noise(5, B, T * hop)
    sd = WT.make_vocoder_state_dict(cfg, seed=6)
    ctx = Context("cuda:0", precision="f32")
    v = Vocoder(ctx, cfg, sd)
    wav = v(mel, f0, rand_ini=rand_ini, noise=noise).cpu()
    with torch.no_grad():
        ref = N.hifigan_nsf_forward(O_voc.fold_weight_norm(sd), cfg, mel, f0, rand_ini, noise)
    check("f32_hifigan_nsf_600_frames_vs_oracle", wav, ref, 2e-4)
    assert float(((wav.double() - ref.double()) ** 2).mean().sqrt()) <= 1e-4
    v.close()
    ctx.close()


def test_hifigan_wrapper_spec2wav_with_f0(golden):
    """NeuralSeq/vocoders/hifigan.py:55-69 with use_nsf: spec2wav(mel [T, 80], f0=f0 [T]) -> wav [T*hop] ndarray; the
    two SineGen draws come from torch's global generator on the device, as in the reference."""
    from audiogpt_amd.vocoder.hifigan import HifiGAN
    from oracle import nsf as N
    from oracle import vocoder as O_voc
    g = golden("hifigan_nsf_24k")
    h = dict(C.HIFIGAN_NSF_24K, audio_sample_rate=24000, use_nsf=True)
    sd = WT.make_vocoder_state_dict(C.HIFIGAN_NSF_24K, seed=6)
    voc = HifiGAN(h, device="cuda:0", state_dict=sd, precision="f32")
    mel, f0 = g["mel"][0], g["f0"][0]
    torch.manual_seed(77)
    wav = voc.spec2wav(mel.T, f0=f0)
    assert isinstance(wav, np.ndarray) and wav.dtype == np.float32 and wav.shape == (40 * 128,)
    torch.manual_seed(77)
    rand_ini = torch.rand(1, 9, device="cuda").cpu()
    noise = torch.randn(1, 40 * 128, 9, device="cuda").cpu()
    with torch.no_grad():
        ref = N.hifigan_nsf_forward(O_voc.fold_weight_norm(sd), C.HIFIGAN_NSF_24K, torch.from_numpy(mel)[None],
                                    torch.from_numpy(f0)[None], rand_ini, noise)
    check("f32_HifiGAN.spec2wav_f0_vs_oracle", wav, ref.reshape(-1), 2e-4)
    # without f0 the same object vocodes through the plain generator path only if it has no NSF branch
    plain = HifiGAN(dict(C.HIFIGAN_NS_128), device="cuda:0", precision="f32")
    assert plain.spec2wav(g["mel"][0].T, f0=f0).shape == (40 * 256,)      # use_nsf off: f0 ignored, as the reference does
