"""16 kHz log-mel front end of the inpainting tool.

Restates `TRANSFORMS_16000` (text_to_audio/Make_An_Audio/ldm/data/extract_mel_spectrogram.py:15-38, 140-150) and
`Inpaint.gen_mel_audio` (audio-chatgpt.py:468-491).  Two implementations of the same arithmetic:
  * `DeviceMelTransform` -- the product path: the framed DFT and the mel filter bank are two GEMMs on the MI355X
    (`maa_spectral_*`, csrc/clap_audio.cpp), so a clip goes waveform -> mel -> VAE encoder -> ... without leaving the GPU;
    this module only builds the two constant matrices (windowed DFT basis, filter bank);
  * `transforms_16000` / `gen_mel_audio` -- the numpy restatement the device path is tested against.

PARITY UNPINNED for the two pieces that live in a third-party dependency absent from this image -- librosa (listed
without a version in the reference's requirements.txt:25):
  * `librosa.stft(x, n_fft=1024, hop_length=256)`: centre padding n_fft // 2 on both sides, periodic Hann window
    (scipy.signal.get_window("hann", 1024, fftbins=True)), one-sided FFT, frame t covers samples
    [t hop - 512, t hop + 512) of the signal, 1 + len(x) // hop frames.  `PAD_MODE`: the padding is "reflect" up to
    librosa 0.9.2 and zeros ("constant") from 0.10.  The reference's environment is 0.9.x -- audio-chatgpt.py:814 calls
    `librosa.resample(y, sr, 22050)` positionally, which 0.10 rejects (keyword-only), and requirements.txt pins
    resampy==0.2.2 (0.9's resampler backend), torchaudio==0.12.1 and numpy==1.23.1 (mid-2022) -- so "reflect" is the
    default; AUDIOGPT_AMD_STFT_PAD=constant (or `pad_mode=`) selects the 0.10 behaviour.  Only the first and last two
    frames of a clip differ between the two;
  * `librosa.filters.mel(sr, n_fft, n_mels=80, fmin=125, fmax=7600)`: Slaney mel scale (linear below 1 kHz with
    200/3 Hz per mel, logarithmic above with step log(6.4) / 27), triangular filters between consecutive mel
    frequencies evaluated at the FFT bin centres, Slaney area normalisation 2 / (f[i+2] - f[i]).
They are restated below from librosa's published algorithm; tests/test_host_logic.py checks them against an independent
scipy.signal.stft framing, analytic tones, the filter bank's defining properties, and against `transformers.audio_utils`
(an independent implementation of the same two librosa routines: filter bank equal to 1e-9, |STFT| to 1e-6 relative) --
not against librosa itself.
`librosa.resample` (resampy "kaiser_best" in 0.9.2) is replaced by scipy.signal.resample_poly when the input is not
already at 16 kHz -- a different (polyphase Kaiser) low-pass; at 16 kHz no resampling happens, as in the reference.
"""
import os

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 1024
HOP = N_FFT // 4
N_MELS = 80
FMIN, FMAX = 125.0, 7600.0
MEL_LEN = 848
PAD_MODE = os.environ.get("AUDIOGPT_AMD_STFT_PAD", "reflect")      # librosa <= 0.9.2 (the reference's); "constant" = librosa >= 0.10
if PAD_MODE not in ("reflect", "constant"):
    raise ValueError("AUDIOGPT_AMD_STFT_PAD must be 'reflect' (librosa <= 0.9.2, the default) or 'constant' (librosa >= 0.10), got %r"
                     % PAD_MODE)


def hz_to_mel(f):
    """Slaney scale (librosa.hz_to_mel, htk=False)."""
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """librosa.filters.mel(..., htk=False, norm="slaney") -> [n_mels, 1 + n_fft // 2] float32."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def stft_magnitude(x, n_fft=N_FFT, hop=HOP, pad_mode=None):
    """|librosa.stft(x, n_fft, hop_length=hop)| for a 1-D float signal -> [1 + n_fft // 2, 1 + len(x) // hop]."""
    from scipy.signal import get_window
    x = np.asarray(x, dtype=np.float32)
    pad_mode = pad_mode or PAD_MODE
    if pad_mode == "reflect" and x.shape[0] <= n_fft // 2:
        raise ValueError("signal too short for reflect padding")
    win = get_window("hann", n_fft, fftbins=True).astype(np.float32)
    xp = np.pad(x, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (xp.shape[0] - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = xp[idx] * win[:, None]
    return np.abs(np.fft.rfft(frames, axis=0)).astype(np.float32)


_MEL_BASIS = None


def transforms_16000(wav):
    """TRANSFORMS_16000: mel spectrogram (power 1) -> max(1e-5, .) -> log10 -> *20 -> -20 -> +100 -> /100 -> clip(0, 1).
    wav: 1-D float in [-1, 1] at 16 kHz -> [80, 1 + len(wav) // 256] in [0, 1]."""
    global _MEL_BASIS
    if _MEL_BASIS is None:
        _MEL_BASIS = mel_filterbank()
    mel = np.dot(_MEL_BASIS, stft_magnitude(wav))
    x = np.log10(np.maximum(1e-5, mel))
    x = (x * 20 - 20 + 100) / 100
    return np.clip(x, 0, 1.0)


def dft_basis(n_fft=N_FFT, window="hann"):
    """The one-sided DFT as a real matrix [2 (n_fft // 2 + 1), n_fft] with the periodic analysis window folded in: rows
    0 .. n_fft/2 = w[n] cos(2 pi k n / N), the next n_fft/2 + 1 rows = -w[n] sin(2 pi k n / N) (the layout of torchlibrosa's
    stft.conv_real.weight / conv_imag.weight; the STFT of a frame is basis @ frame)."""
    from scipy.signal import get_window
    win = get_window(window, n_fft, fftbins=True).astype(np.float64)
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    n = np.arange(n_fft, dtype=np.float64)[None, :]
    # k n mod N keeps the argument of cos / sin small (exact in fp64 for these sizes)
    ang = 2.0 * np.pi * np.mod(k * n, n_fft) / n_fft
    return np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], axis=0).astype(np.float32)


def prepare_wav(input_audio):
    """The host half of Inpaint.gen_mel_audio (audio-chatgpt.py:473-489): (sr, int16 samples, mono or [n, 2] stereo) ->
    float32 mono at 16 kHz, zero-extended / cropped to the clip length."""
    sr, wav = input_audio
    wav = np.asarray(wav).astype(np.float32, order="C") / 32768.0
    if wav.ndim == 2:
        wav = wav.mean(axis=1)                  # librosa.to_mono
    if sr != SAMPLE_RATE:
        from math import gcd

        from scipy.signal import resample_poly
        g = gcd(int(sr), SAMPLE_RATE)
        wav = resample_poly(wav, SAMPLE_RATE // g, int(sr) // g).astype(np.float32)
    input_len = MEL_LEN * HOP
    if len(wav) < input_len:
        wav = np.pad(wav, (0, input_len), constant_values=0)
    else:
        wav = wav[:input_len]
    return wav


def gen_mel_audio(input_audio):
    """Inpaint.gen_mel_audio (audio-chatgpt.py:468-491) in numpy: -> [80, 849] mel of the first 848 * 256 samples
    (zero-extended when shorter; the reference pads by a full clip length, so the frame count then depends on the
    input -- reproduced)."""
    return transforms_16000(prepare_wav(input_audio))


class DeviceMelTransform:
    """TRANSFORMS_16000 on the MI355X: `(sr, wav) -> [80, frames]` like `gen_mel_audio`, the STFT and the filter bank as
    two GEMMs of the library (exact fp32).  `ctx`: the backend context of the model the mel is for."""

    def __init__(self, ctx, pad_mode=None):
        from .backend import Spectral
        self.pad_mode = pad_mode or PAD_MODE
        cfg = dict(n_fft=N_FFT, hop=HOP, n_mels=N_MELS, pad_mode=self.pad_mode, power=1, log_kind="transforms_16000",
                   amin=1e-5, ref=1.0, out_layout="bmt")
        self.spectral = Spectral(ctx, cfg, dft_basis(N_FFT), mel_filterbank())

    def mel(self, wav):
        """float waveform [n] or [B, n] at 16 kHz -> device tensor [B, 80, 1 + n // 256] in [0, 1]."""
        return self.spectral.forward(wav)

    def __call__(self, sr, wav):
        import torch
        x = torch.from_numpy(np.ascontiguousarray(prepare_wav((sr, wav))))
        return self.mel(x)[0].cpu().numpy()
