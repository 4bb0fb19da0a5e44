"""Oracle: the CLAP best-of-n scorer (T2A.select_best_audio, audio-chatgpt.py:185-199), functional CPU fp32.
TEST INFRASTRUCTURE ONLY -- the product is audiogpt_amd/clap.py over csrc/clap_audio.cpp.

Restates:
  /root/reference/text_to_audio/Make_An_Audio/ldm/modules/encoders/CLAP/audio.py:14-47 (ConvBlock: 3x3 conv without bias
      -> BatchNorm -> ReLU, twice, then average pooling), :113-179 (Cnn14.forward after its two extractors: bn0 over the
      mel bins, six blocks, mean over frequency, max + mean over time, relu(fc1))
  .../CLAP/clap.py:8-20 (Projection), :22-39 (AudioEncoder: projection of the 2048-d embedding)
  .../wav_evaluation/models/CLAPWrapper.py:186-191, 207-215 (unit-length embeddings, similarity = text @ audio^T)
  .../wav_evaluation/models/clap.py:41-53 (TextEncoder: BERT under the tokenizer's attention mask, [CLS] row, Projection)
  .../CLAPWrapper.py:103-128 (resample_and_duration)
The waveform front end lives in two third-party packages that are absent here; it is restated from their published
algorithms -- PARITY UNPINNED against the packages themselves:
  torchaudio 0.12.1 functional.resample (sinc_interpolation, lowpass_filter_width 6, rolloff 0.99): `resample` below;
  torchlibrosa Spectrogram (conv1d DFT, Hann, centre + reflect padding, power 2) and LogmelFilterBank (librosa mel
  filters, 10 log10(clamp(., 1e-10)), top_db None): `logmel` below, through torch.stft (the same transform by
  definition) and the slaney filter bank of audiogpt_amd.mel (pinned to transformers.audio_utils in test_host_logic).
"""
import math

import torch
import torch.nn.functional as F


def _bn(x, sd, p, eps=1e-5):
    """BatchNorm in eval mode over dim 1."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - sd[p + ".running_mean"].view(shape)) / torch.sqrt(sd[p + ".running_var"].view(shape) + eps) * \
        sd[p + ".weight"].view(shape) + sd[p + ".bias"].view(shape)


def cnn14_embedding(sd, cfg, logmel):
    """Cnn14.forward from the log-mel on (audio.py:150-176, eval mode): [B, 1, T, mel_bins] -> [B, out_emb]."""
    x = _bn(logmel.transpose(1, 3), sd, "base.bn0").transpose(1, 3)                 # (:150-152)
    n = len(cfg["channels"])
    for i in range(n):
        p = "base.conv_block%d." % (i + 1)
        x = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], padding=1), sd, p + "bn1"))
        x = F.relu(_bn(F.conv2d(x, sd[p + "conv2.weight"], padding=1), sd, p + "bn2"))
        if i < n - 1:
            x = F.avg_pool2d(x, kernel_size=2)                                      # the last block pools (1, 1)
    x = x.mean(dim=3)                                                               # (:166)
    x = x.max(dim=2)[0] + x.mean(dim=2)                                             # (:168-170)
    return F.relu(F.linear(x, sd["base.fc1.weight"], sd["base.fc1.bias"]))          # (:172-173)


def clap_audio_embed(sd, cfg, logmel):
    """AudioEncoder.forward + CLAPWrapper's normalisation: -> [B, d_proj], unit length."""
    e = cnn14_embedding(sd, cfg, logmel)
    e1 = F.linear(e, sd["projection.linear1.weight"])
    e2 = F.linear(F.gelu(e1), sd["projection.linear2.weight"])
    D = e1.shape[-1]
    z = F.layer_norm(e1 + e2, (D,), sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"], 1e-5)
    return z / z.norm(dim=-1, keepdim=True)


def similarity(audio_embeddings, text_embeddings):
    """CLAPWrapper.compute_similarity(use_logit_scale=False) (:207-215): [n_audio, n_text]."""
    return (text_embeddings @ audio_embeddings.T).T


def text_embedding(sd, cfg, ids):
    """TextEncoder.forward + CLAPWrapper's normalisation for ONE unpadded id row [n]: under the attention mask the padded
    keys have zero weight, so BERT on the real tokens alone gives the same [CLS] row (checked against the reference's
    padded run in tests/test_oracle_golden.py).  sd: `caption_encoder.`-relative keys."""
    from . import encoders
    h = encoders.bert_forward(sd, cfg, ids.reshape(1, -1))[:, 0]
    e1 = F.linear(h, sd["projection.linear1.weight"])
    e2 = F.linear(F.gelu(e1), sd["projection.linear2.weight"])
    D = e1.shape[-1]
    z = F.layer_norm(e1 + e2, (D,), sd["projection.layer_norm.weight"], sd["projection.layer_norm.bias"], 1e-5)
    return z / z.norm(dim=-1, keepdim=True)


def resample(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample, default method: x [B, n] -> [B, ceil(new n / orig)]."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base / orig)
    kernels = kernels.to(torch.float32)
    n = x.shape[-1]
    y = F.conv1d(F.pad(x, (width, width + orig))[:, None], kernels, stride=orig)       # [B, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    return y[:, :math.ceil(new * n / orig)]


def logmel(x, melw, n_fft=1024, hop=320, amin=1e-10, ref=1.0):
    """Cnn14's two extractors: x [B, n] -> [B, 1, frames, mel_bins]; melw [n_freq, mel_bins]."""
    win = torch.hann_window(n_fft, periodic=True)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=win, center=True, pad_mode="reflect",
                      return_complex=True)                                              # [B, n_freq, frames]
    power = (spec.real ** 2 + spec.imag ** 2).transpose(1, 2)
    mel = power @ melw
    return (10.0 * torch.log10(torch.clamp(mel, min=amin)) - 10.0 * math.log10(max(amin, ref)))[:, None]


def resample_and_duration(wav, sample_rate, duration, target_rate, start):
    """CLAPWrapper.resample_and_duration with the crop position given: 1-D wav -> duration * sample_rate samples."""
    x = resample(wav.reshape(1, -1), sample_rate, target_rate).reshape(-1)
    want = duration * sample_rate
    if want >= x.shape[0]:
        return x.repeat(math.ceil(want / x.shape[0]))[:want]
    return x[start:start + want]
