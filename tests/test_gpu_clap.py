"""SURVEY 8f / N4 on the device: the CLAP best-of-n scorer (csrc/clap_audio.cpp, spectral.hip, audiogpt_amd/clap.py) and the
16 kHz mel front end of the inpainting tool (audiogpt_amd/mel.DeviceMelTransform).

Goldens from the reference's own classes (tests/golden/make_golden.py clapaudio | clapscore): `clap_audio_cnn14` = the CLAP
AudioEncoder (Cnn14 + Projection) from the log-mel on; `clap_score` = the scorer through wav_evaluation's TextEncoder (padded
prompt + attention mask), AudioEncoder and CLAPWrapper's normalisation / similarity.  The waveform front end (torchaudio
resampler, torchlibrosa extractors: both absent) is compared with the oracle's restatement (oracle/clap_audio.py).
"""
import numpy as np
import pytest
import torch

from audiogpt_amd import config as C
from audiogpt_amd import weights as WT
from tests.util import check

pytestmark = pytest.mark.gpu


def _score_logmel(g, cfg):
    frames = int(g["frames"])
    sc, of, ti = (torch.from_numpy(g[k]).view(-1, 1, 1, 1) for k in ("scales", "offsets", "tilts"))
    return torch.randn(3, 1, frames, cfg["mel_bins"], generator=torch.Generator().manual_seed(int(g["logmel_seed"]))) * sc + \
        of + ti * torch.arange(cfg["mel_bins"]).view(1, 1, 1, -1)


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("bf16x3", 1e-4)])
def test_clap_audio_branch_matches_reference(golden, precision, tol):
    from audiogpt_amd.backend import ClapAudio, Context
    g = golden("clap_audio_cnn14")
    cfg = C.CLAP_AUDIO_CNN14
    ctx = Context("cuda:0", precision=precision)
    enc = ClapAudio(ctx, cfg, WT.make_clap_audio_state_dict(cfg, seed=14))
    logmel = torch.randn(2, 1, cfg["frames"], cfg["mel_bins"], generator=torch.Generator().manual_seed(int(g["logmel_seed"]))) * 12.0 - 30.0
    z, emb = enc.embed(logmel, return_embedding=True)
    check(f"{precision}_clap_audio_embedding_vs_reference", emb, g["embedding"], tol)
    check(f"{precision}_clap_audio_z_vs_reference", z, g["z"], tol)
    # a clip's embedding does not depend on the batch it was embedded in
    z1 = enc.embed(logmel[1:])
    assert torch.equal(z1, z[1:])


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_clap_score_matches_reference(golden, precision):
    """Text [CLS] embedding from the UNPADDED ids == the reference's padded + masked run; audio embeddings; similarity and
    the index select_best_audio would pick."""
    from audiogpt_amd.clap import CLAPWrapper
    g = golden("clap_score")
    cfg = C.CLAP_SCORER
    clap = CLAPWrapper(device="cuda:0", precision=precision, synthetic=True)      # seeded weights (21 / 22), as the golden
    te = clap.get_text_embeddings_from_ids([g["input_ids"]])
    check(f"{precision}_clap_score_text_vs_reference", te, g["text_embedding"], 1e-4)
    ae = clap.audio_encoder.embed(_score_logmel(g, cfg))
    check(f"{precision}_clap_score_audio_vs_reference", ae, g["audio_embedding"], 1e-4)
    sim = clap.compute_similarity(ae, te, use_logit_scale=False)
    assert sim.shape == (3, 1)
    assert np.abs(sim.cpu().numpy() - g["similarity"]).max() < 2e-5
    assert int(sim.argmax()) == int(g["similarity"].argmax())
    scaled = clap.compute_similarity(ae, te)
    assert torch.allclose(scaled, sim / 0.07, rtol=1e-5)

    class Tok:                      # stands in for AutoTokenizer.encode_plus: pads to text_len and returns the mask
        def encode_plus(self, text, add_special_tokens, max_length, padding, return_tensors):
            assert max_length == 100 and padding == "max_length" and add_special_tokens
            ids = torch.zeros(1, max_length, dtype=torch.long)
            n = len(g["input_ids"])
            ids[0, :n] = torch.from_numpy(g["input_ids"])
            mask = torch.zeros_like(ids)
            mask[0, :n] = 1
            return {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}
    clap.tokenizer = Tok()
    assert torch.equal(clap.get_text_embeddings(["a dog barking in the rain"]), te)


def test_front_ends_match_the_oracle():
    """Resampler, 44.1 kHz log-mel and the 16 kHz TRANSFORMS_16000 on the device against the CPU restatements."""
    from audiogpt_amd import mel as M
    from audiogpt_amd.backend import Context, Resampler, Spectral
    from audiogpt_amd.clap import sinc_resample_kernel
    from oracle import clap_audio as O
    ctx = Context("cuda:0", precision="bf16x3")          # the front ends run in exact fp32 whatever the context says
    rs = np.random.RandomState(3)
    t = np.arange(2 * 16000) / 16000.0
    wav = (0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.1 * rs.randn(len(t))).astype(np.float32)
    x = torch.from_numpy(np.stack([wav, wav[::-1].copy()]))
    k, width = sinc_resample_kernel(16000, 44100)
    r = Resampler(ctx, 160, 441, width, k)
    y = r(x)
    ref = O.resample(x, 16000, 44100)
    assert y.shape == ref.shape == (2, 88200)
    check("resampler_vs_oracle", y, ref, 1e-5)
    a = C.CLAP_SCORER
    melw = M.mel_filterbank(sr=a["sampling_rate"], n_fft=1024, n_mels=64, fmin=a["fmin"], fmax=a["fmax"])
    sp = Spectral(ctx, dict(n_fft=1024, hop=320, n_mels=64, pad_mode="reflect", power=2, log_kind="db", amin=1e-10, ref=1.0,
                            out_layout="btm"), M.dft_basis(1024), melw)
    lm = sp(ref)
    lref = O.logmel(ref, torch.from_numpy(melw.T.copy()))[:, 0]
    assert lm.shape == lref.shape == (2, 1 + 88200 // 320, 64)
    d = (lm.cpu() - lref).abs()
    assert float(d.max()) < 5e-3 and float(d.mean()) < 2e-4, (float(d.max()), float(d.mean()))     # dB
    # the inpainting tool's mel: both padding conventions, a full-length clip and a short one (zero-extended)
    for pad_mode in ("reflect", "constant"):
        tr = M.DeviceMelTransform(ctx, pad_mode=pad_mode)
        for n in (848 * 256 + 1000, 30000):
            clip = (rs.randn(n) * 3000).astype(np.int16)
            got = tr(16000, clip)
            M.PAD_MODE, saved = pad_mode, M.PAD_MODE
            try:
                want = M.gen_mel_audio((16000, clip))
            finally:
                M.PAD_MODE = saved
            assert got.shape == want.shape and got.shape[0] == 80
            assert np.abs(got - want).max() <= 1e-5, (pad_mode, n, np.abs(got - want).max())
    stereo = (rs.randn(40000, 2) * 2000).astype(np.int16)
    assert np.abs(M.DeviceMelTransform(ctx)(16000, stereo) - M.gen_mel_audio((16000, stereo))).max() <= 1e-5


def test_clap_wrapper_scores_waveforms_like_the_oracle_chain():
    """The whole scorer from 16 kHz waveforms: resample -> crop (pinned start) -> log-mel -> Cnn14 -> similarity against the
    oracle chain, and T2A.select_best_audio's choice."""
    from audiogpt_amd import mel as M
    from audiogpt_amd.clap import CLAPWrapper
    from oracle import clap_audio as O
    cfg = C.CLAP_SCORER
    clap = CLAPWrapper(device="cuda:0", precision="bf16x3", crop_start=12345, synthetic=True)
    tsd = WT.make_clap_text_state_dict(cfg["text"], seed=21)
    asd = WT.make_clap_audio_state_dict(cfg["audio"], seed=22)
    rs = np.random.RandomState(5)
    n = 159744                                                        # a 10-s Make-An-Audio clip: 624 frames * 256
    t = np.arange(n) / 16000.0
    wavs = [(0.2 * np.sin(2 * np.pi * f * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + s * rs.randn(n)).astype(np.float32)
            for f, s in ((300.0, 0.02), (1200.0, 0.1), (3500.0, 0.3))]
    ids = torch.tensor([101, 2023, 2003, 1037, 3231, 102])
    melw = torch.from_numpy(M.mel_filterbank(sr=44100, n_fft=1024, n_mels=64, fmin=50, fmax=14000).T.copy())
    with torch.no_grad():
        te = O.text_embedding(tsd, cfg["text"], ids)
        ref = []
        for w in wavs:
            clip = O.resample_and_duration(torch.from_numpy(w), 16000, cfg["duration"], 44100, 12345)
            assert clip.shape[0] == 9 * 16000
            z = O.clap_audio_embed(asd, cfg["audio"], O.logmel(clip[None], melw))
            ref.append(float(O.similarity(z, te)))
    te_d = clap.get_text_embeddings_from_ids([ids])
    got = [float(clap.compute_similarity(clap.get_audio_embeddings([(torch.FloatTensor(w), 16000)], resample=True), te_d,
                                         use_logit_scale=False)) for w in wavs]
    assert np.abs(np.array(got) - np.array(ref)).max() < 5e-5, (got, ref)
    assert int(np.argmax(got)) == int(np.argmax(ref))
    # a clip shorter than duration * sr is repeated (CLAPWrapper.py:113-119)
    short = clap.resample_and_duration((torch.from_numpy(wavs[0][:20000]), 16000), cfg["duration"], resample=True)
    assert short.shape[0] == 9 * 16000


def test_T2A_select_best_audio_uses_the_device_scorer():
    from audiogpt_amd.clap import CLAPWrapper
    from audiogpt_amd.tools import T2A

    class Tok:
        def __call__(self, text):
            return [101, 2023, 2003, 1037, 3231, 102]
    t2a = T2A.__new__(T2A)                     # select_best_audio only needs the scorer fields
    t2a.scorer = None
    t2a.clap_model = CLAPWrapper(device="cuda:0", precision="bf16x3", tokenizer=Tok(), crop_start=777, synthetic=True)
    rs = np.random.RandomState(7)
    n = 159744
    t = np.arange(n) / 16000.0
    wav_list = [(16000, (0.2 * np.sin(2 * np.pi * f * t) + 0.05 * rs.randn(n)).astype(np.float32)) for f in (250.0, 900.0, 4000.0)]
    best = t2a.select_best_audio("this is a test", wav_list)
    te = t2a.clap_model.get_text_embeddings(["this is a test"])
    scores = [float(t2a.clap_model.compute_similarity(t2a.clap_model.get_audio_embeddings([(torch.FloatTensor(w), sr)], resample=True),
                                                      te, use_logit_scale=False)) for sr, w in wav_list]
    assert best is wav_list[int(np.argmax(scores))]
    t2a.clap_model = None
    assert t2a.select_best_audio("x", wav_list) is wav_list[0]
    t2a.scorer = lambda prompt, wav, sr: float(np.abs(wav).sum())
    assert t2a.select_best_audio("x", wav_list) is max(wav_list, key=lambda p: np.abs(p[1]).sum())
