"""Oracle: the audio branch of the CLAP best-of-n scorer from the log-mel on, functional CPU fp32.  TEST INFRASTRUCTURE ONLY.
Groundwork for the rest of SURVEY 8f / N4 (T2A.select_best_audio, audio-chatgpt.py:185-199) -- no HIP implementation yet.

Restates:
  /root/reference/text_to_audio/Make_An_Audio/ldm/modules/encoders/CLAP/audio.py:14-47 (ConvBlock: 3x3 conv without bias
      -> BatchNorm -> ReLU, twice, then average pooling), :113-179 (Cnn14.forward after its two extractors: bn0 over the
      mel bins, six blocks, mean over frequency, max + mean over time, relu(fc1))
  .../CLAP/clap.py:8-20 (Projection), :22-39 (AudioEncoder: projection of the 2048-d embedding)
  .../wav_evaluation/models/CLAPWrapper.py:186-191, 207-215 (unit-length embeddings, similarity = text @ audio^T)
Not restated: the waveform front end (torchaudio resampling to 44.1 kHz, the random 5-s crop, torchlibrosa's Spectrogram /
LogmelFilterBank) -- those dependencies are absent here, so the oracle starts at the [B, 1, frames, 64] log-mel.
"""
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
