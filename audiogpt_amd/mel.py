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
`librosa.resample(y, orig_sr=sr, target_sr=16000)` (audio-chatgpt.py:462, 482; librosa 0.9.x: res_type "kaiser_best" =
resampy 0.2.2, pinned in requirements.txt:52) is restated too -- `resampy_filter`, `resampy_kernel_bank`, `librosa_resample`:
resampy's band-limited sinc interpolation (J. O. Smith's algorithm) with its published "kaiser_best" design (64 zero crossings,
512 table entries per crossing, Kaiser window beta = 14.769656459379492, roll-off 0.9475937167399596), the table linearly
interpolated at each tap, zero extension at both ends, int(n ratio) outputs, then librosa's fix_length to ceil(n ratio) (a zero
sample when the two differ) and no rescaling.  For a rational ratio the interpolated taps repeat with the period new / gcd, so
the whole routine is a bank of `new` FIR phases applied every `orig` input samples -- the strided contraction the library's
resampler (`maa_resampler_*`) runs; `DeviceMelTransform` feeds it this bank.  tests/test_host_logic.py checks the bank against
the scalar time-register loop of oracle/resampy.py (the algorithm as resampy states it), tones, and scipy's polyphase resampler
(a different Kaiser low-pass: agreement to the pass-band ripple, stated there); resampy itself is absent: PARITY UNPINNED.
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


RESAMPY_FILTERS = {      # resampy 0.2.2 filters.py: the published designs behind data/kaiser_best.npz / kaiser_fast.npz
    "kaiser_best": dict(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596),
    "kaiser_fast": dict(num_zeros=16, precision=9, beta=8.555504641634386, rolloff=0.85),
}
_RESAMPY_CACHE = {}


def resampy_filter(name="kaiser_best"):
    """resampy.filters.sinc_window(num_zeros, precision, window=kaiser(beta), rolloff): the right half of the windowed sinc
    sampled 2**precision times per zero crossing -> (half_window float64 [num_zeros 2**precision + 1], 2**precision)."""
    if name not in _RESAMPY_CACHE:
        from scipy.signal.windows import kaiser
        f = RESAMPY_FILTERS[name]
        num_bits = 2 ** f["precision"]
        n = num_bits * f["num_zeros"]
        sinc_win = f["rolloff"] * np.sinc(f["rolloff"] * np.linspace(0, f["num_zeros"], num=n + 1, endpoint=True))
        taper = kaiser(2 * n + 1, f["beta"])[n:]
        _RESAMPY_CACHE[name] = (taper * sinc_win, num_bits)
    return _RESAMPY_CACHE[name]


def resampy_kernel_bank(sr_orig, sr_new, filter="kaiser_best"):
    """resampy.resample(x, sr_orig, sr_new, filter) as a polyphase bank: -> (kernels float32 [new, 2 width + orig], width)
    with orig / new divided by their gcd, in the layout of `clap.sinc_resample_kernel` (output q new + p is the dot product
    of kernels[p] with the input samples q orig - width ... q orig + orig + width - 1, zeros outside the signal).  Phase p
    sits at time p orig / new: n = floor, frac = scale (time - n); left-wing tap i multiplies x[n - i] by the table at
    frac 2**precision + i index_step, right-wing tap k multiplies x[n + 1 + k] by the table at (scale - frac) 2**precision +
    k index_step, each linearly interpolated between table entries (resampy/interpn.py:resample_f)."""
    from math import gcd
    g = gcd(int(sr_orig), int(sr_new))
    orig, new = int(sr_orig) // g, int(sr_new) // g
    win, num_table = resampy_filter(filter)
    sample_ratio = float(sr_new) / float(sr_orig)
    win = win * sample_ratio if sample_ratio < 1 else win.copy()
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, sample_ratio)
    index_step = int(scale * num_table)
    nwin = win.shape[0]
    width = nwin // index_step + 1
    bank = np.zeros((new, 2 * width + orig), dtype=np.float64)
    for p in range(new):
        time = p * orig / new
        n = (p * orig) // new
        frac = scale * (time - n)
        for wing, f in ((0, frac), (1, scale - frac)):
            index_frac = f * num_table
            offset = int(index_frac)
            eta = index_frac - offset
            idx = offset + index_step * np.arange((nwin - offset) // index_step)
            w = win[idx] + eta * delta[idx]
            taps = np.arange(idx.shape[0])
            cols = width + (n - taps if wing == 0 else n + 1 + taps)
            bank[p, cols] += w
    return bank.astype(np.float32), width


def resampy_carry_kernel(sr_orig, sr_new, filter="kaiser_best"):
    """The one place where resampy is NOT periodic in the output index.  Its time register is a running float64 sum of
    1 / ratio; where the exact time is an integer N (outputs q new) the sum may land a few ulps below it, and then n = N - 1 and
    frac = scale (1 - eps) instead of n = N, frac = 0.  With an ideal table the two are the same point of a continuous filter,
    but resampy steps through its table by the TRUNCATED index_step = int(scale 2**precision), so the left wing then sits a
    fraction of a table entry further out (a 1e-4-level difference).  -> the kernel [1, 2 width + orig] of that case in the
    window layout of `resampy_kernel_bank`; `resampy_carries` says which outputs take it (none when 1 / ratio is exact in
    binary: 48 / 32 / 24 / 8 kHz)."""
    from math import gcd
    g = gcd(int(sr_orig), int(sr_new))
    orig, new = int(sr_orig) // g, int(sr_new) // g
    sample_ratio = float(sr_new) / float(sr_orig)
    win, num_table = resampy_filter(filter)
    win = win * sample_ratio if sample_ratio < 1 else win.copy()
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, sample_ratio)
    index_step = int(scale * num_table)
    nwin = win.shape[0]
    width = nwin // index_step + 1
    k = np.zeros((1, 2 * width + orig), dtype=np.float64)
    index_frac = scale * num_table                                  # left wing at frac -> scale: x[N - 1 - i]
    offset = int(index_frac)
    idx = offset + index_step * np.arange((nwin - offset) // index_step)
    k[0, width - 1 - np.arange(idx.shape[0])] += win[idx] + (index_frac - offset) * delta[idx]
    idx = index_step * np.arange(nwin // index_step)                # right wing at frac -> 0: x[N + k]
    k[0, width + np.arange(idx.shape[0])] += win[idx]
    return k.astype(np.float32)


def resampy_carries(n_out, sr_orig, sr_new):
    """Output indices t (multiples of new / gcd) at which resampy's running time register int()s to one less than the
    exact sample index (see `resampy_carry_kernel`)."""
    from math import gcd
    g = gcd(int(sr_orig), int(sr_new))
    orig, new = int(sr_orig) // g, int(sr_new) // g
    inc = 1.0 / (float(sr_new) / float(sr_orig))
    if n_out <= new:
        return np.zeros(0, dtype=np.int64)
    reg = np.cumsum(np.full(n_out - 1, inc))                        # reg[t - 1] = the register before output t (sequential adds)
    t = np.arange(new, n_out, new)
    return t[reg[t - 1].astype(np.int64) != (t // new) * orig]


def librosa_resample(wav, sr_orig, sr_new=SAMPLE_RATE, filter="kaiser_best"):
    """librosa.resample(wav, orig_sr, target_sr) of librosa 0.9.x (res_type "kaiser_best", fix=True, scale=False) for a 1-D
    float32 signal, in numpy: the kernel bank applied as a strided contraction (the arithmetic of the device path)."""
    from math import gcd
    wav = np.asarray(wav, dtype=np.float32)
    if int(sr_orig) == int(sr_new):
        return wav
    g = gcd(int(sr_orig), int(sr_new))
    orig, new = int(sr_orig) // g, int(sr_new) // g
    k, width = resampy_kernel_bank(sr_orig, sr_new, filter)
    ratio = float(sr_new) / float(sr_orig)
    n_resampy, n_fixed = int(wav.shape[0] * ratio), int(np.ceil(wav.shape[0] * ratio))
    q = -(-n_fixed // new)
    xp = np.zeros(width + q * orig + orig + width, dtype=np.float32)
    m = min(wav.shape[0], q * orig + orig + width)
    xp[width:width + m] = wav[:m]
    frames = np.lib.stride_tricks.sliding_window_view(xp, k.shape[1])[::orig][:q]
    out = (frames @ k.T).reshape(-1)[:n_fixed].astype(np.float32)
    t = resampy_carries(n_resampy, sr_orig, sr_new)
    if t.size:
        out[t] = frames[t // new] @ resampy_carry_kernel(sr_orig, sr_new, filter)[0]
    out[n_resampy:] = 0                          # resampy stops at int(n ratio); librosa.util.fix_length zero-pads to ceil
    return out


def to_float_mono(wav):
    """audio-chatgpt.py:457-460 / 473-480: int16 samples -> float32 / 32768, stereo [n, 2] -> librosa.to_mono (channel mean)."""
    wav = np.asarray(wav).astype(np.float32, order="C") / 32768.0
    if wav.ndim == 2:
        wav = wav.mean(axis=1)                  # librosa.to_mono(wav.T)
    return wav


def fit_clip(wav):
    """audio-chatgpt.py:464-469 / 484-489: crop to 848 * 256 samples; a shorter clip is zero-extended by a FULL clip length
    (the reference's `np.pad(ori_wav, (0, mel_len * hop_size))`), so its frame count depends on the input -- reproduced."""
    input_len = MEL_LEN * HOP
    if len(wav) < input_len:
        return np.pad(wav, (0, input_len), constant_values=0)
    return wav[:input_len]


def prepare_wav(input_audio):
    """The host restatement of Inpaint.gen_mel_audio up to the mel (audio-chatgpt.py:473-489): (sr, int16 samples, mono or
    [n, 2] stereo) -> float32 mono at 16 kHz, zero-extended / cropped to the clip length."""
    sr, wav = input_audio
    return fit_clip(librosa_resample(to_float_mono(wav), sr, SAMPLE_RATE))


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
        self._resamplers = {}

    def mel(self, wav):
        """float waveform [n] or [B, n] at 16 kHz -> device tensor [B, 80, 1 + n // 256] in [0, 1]."""
        return self.spectral.forward(wav)

    def resample(self, x, sr):
        """librosa.resample(x, orig_sr=sr, target_sr=16000) on the device: resampy's kaiser_best as the library's polyphase
        contraction (`resampy_kernel_bank`), the last sample zeroed where resampy's int(n ratio) falls short of librosa's
        ceil(n ratio).  x: device tensor [n]."""
        from math import gcd

        import torch

        from .backend import Resampler
        sr = int(sr)
        if sr == SAMPLE_RATE:
            return x
        g = gcd(sr, SAMPLE_RATE)
        orig, new = sr // g, SAMPLE_RATE // g
        r = self._resamplers.get(sr)
        if r is None:
            k, width = resampy_kernel_bank(sr, SAMPLE_RATE)
            r = self._resamplers[sr] = Resampler(self.spectral.ctx, orig, new, width, k)
        n = x.shape[-1]
        x = x.contiguous()
        y = r.forward(x)[0]
        ratio = float(SAMPLE_RATE) / float(sr)
        n_resampy, n_fixed = int(n * ratio), int(np.ceil(n * ratio))
        y = y[:n_fixed]
        t = resampy_carries(n_resampy, sr, SAMPLE_RATE)
        if t.size:                               # the outputs whose running time register fell below its integer: a second,
            rc = self._resamplers.get((sr, "carry"))    # one-phase pass of the same contraction, scattered into place
            if rc is None:
                rc = self._resamplers[(sr, "carry")] = Resampler(self.spectral.ctx, orig, 1, r.width,
                                                                 resampy_carry_kernel(sr, SAMPLE_RATE))
            t = torch.from_numpy(t).to(y.device)
            y[t] = rc.forward(x)[0][t // new]
        if n_resampy < n_fixed:
            y[n_resampy:] = 0
        return y

    def __call__(self, sr, wav):
        """Inpaint.gen_mel_audio (audio-chatgpt.py:468-491): (sr, int16 samples [n] or [n, 2]) -> numpy [80, frames]; the
        resampler, the clip fit and the mel all run on the device."""
        import torch
        x = torch.from_numpy(np.ascontiguousarray(to_float_mono(wav))).to(self.spectral.ctx.device)
        x = self.resample(x, sr)
        input_len = MEL_LEN * HOP
        x = torch.nn.functional.pad(x, (0, input_len)) if x.shape[0] < input_len else x[:input_len]
        return self.mel(x.contiguous())[0].cpu().numpy()
