"""16 kHz log-mel front end of the inpainting tool, on the host.

Restates `TRANSFORMS_16000` (text_to_audio/Make_An_Audio/ldm/data/extract_mel_spectrogram.py:15-38, 140-150) and
`Inpaint.gen_mel_audio` (audio-chatgpt.py:468-491).  It runs once per request on a 13.6 s clip (a 1024-point STFT of 848
frames and an 80 x 513 matrix product: ~2 ms of numpy), so it stays on the CPU; the GPU path starts at the VAE encoder.

PARITY UNPINNED for the two pieces that live in a third-party dependency absent from this image -- librosa (listed
without a version in the reference's requirements.txt:25; its call style `librosa.filters.mel(sr=..., n_fft=...)` with
keyword arguments is the 0.10 API, whose defaults are restated here):
  * `librosa.stft(x, n_fft=1024, hop_length=256)`: centre padding n_fft // 2 on both sides with pad_mode="constant"
    (zeros: the default since 0.10; it was "reflect" up to 0.9.2 -- `PAD_MODE` below switches), periodic Hann window
    (scipy.signal.get_window("hann", 1024, fftbins=True)), one-sided FFT, frame t covers samples
    [t hop - 512, t hop + 512) of the signal, 1 + len(x) // hop frames;
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
import numpy as np

SAMPLE_RATE = 16000
N_FFT = 1024
HOP = N_FFT // 4
N_MELS = 80
FMIN, FMAX = 125.0, 7600.0
MEL_LEN = 848
PAD_MODE = "constant"       # librosa >= 0.10; "reflect" reproduces librosa <= 0.9.2


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


def gen_mel_audio(input_audio):
    """Inpaint.gen_mel_audio (audio-chatgpt.py:468-491): (sr, int16 samples, mono or [n, 2] stereo) -> [80, 849] mel of
    the first 848 * 256 samples (zero-extended when shorter; the reference pads by a full clip length, so the frame count
    then depends on the input -- reproduced)."""
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
    return transforms_16000(wav)
