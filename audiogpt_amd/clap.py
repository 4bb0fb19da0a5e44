"""CLAP best-of-n scorer on the MI355X, with the reference's class and method names.

Replaces `wav_evaluation/models/CLAPWrapper.py` (CLAPWrapper: get_text_embeddings :177-182, get_audio_embeddings :184-189,
resample_and_duration :103-128, preprocess_audio :136-150, preprocess_text :152-162, compute_similarity :207-215,
cal_clap_score :217-221) as `T2A.select_best_audio` uses it (audio-chatgpt.py:185-199).  Everything from the 16 kHz
waveform on runs on the device through the C ABI:

  resample 16 k -> 44.1 k   maa_resampler_*   torchaudio.transforms.Resample's sinc kernel bank as one strided contraction
  crop / repeat             tensor slicing (reference quirk kept: the clip is `duration * sample_rate` samples long with
                            sample_rate the INPUT rate, i.e. 9 * 16000 samples of the 44.1 kHz signal)
  log-mel                   maa_spectral_*    torchlibrosa Spectrogram (1024 / hop 320, reflect, power 2) + LogmelFilterBank
  Cnn14 + Projection        maa_clap_audio_*  wav_evaluation/models/audio.py:107-180, clap.py:8-39
  text                      maa_encoder_text_cls   BERT [CLS] -> Projection (clap.py:41-53)
  similarity                maa_clap_similarity

Host side, as in the reference: the tokenizer (`AutoTokenizer.from_pretrained('bert-base-uncased')` needs its vocabulary
file -- a constructor argument here) and the random crop (`random.randrange`, unseeded upstream; `crop_start=` pins it).

PARITY of the two third-party front-end pieces, both absent from this image: torchaudio's resampler and torchlibrosa's
extractors are restated from their published algorithms (`sinc_resample_kernel`, `audiogpt_amd.mel.dft_basis` /
`mel_filterbank`); the spectrogram is pinned to torch.stft and the filter bank to transformers.audio_utils in
tests/test_host_logic.py, the resampler only to its defining properties (unpinned).  When a real CLAP checkpoint is
loaded its own frozen extractor tables (`audio_encoder.base.spectrogram_extractor.stft.conv_{real,imag}.weight`,
`audio_encoder.base.logmel_extractor.melW`) are used instead of the restated ones.
"""
import math
import random

import numpy as np
import torch

from . import config as C
from . import weights as WT
from .backend import ClapAudio, Context, Encoder, Resampler, Spectral, clap_similarity, default_precision
from ._lib import MaaError
from .mel import dft_basis, mel_filterbank


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample's kernel bank for the default "sinc_interpolation" (Hann-windowed sinc), computed
    in float64 and rounded once like torchaudio does: -> (kernels [new, 2 width + orig] float32, width) with orig / new
    divided by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base_freq))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    with np.errstate(divide="ignore", invalid="ignore"):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    return (kernels * window * scale).astype(np.float32), width


def _strip(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


class CLAPWrapper:
    """A class for interfacing the CLAP model (signature of wav_evaluation/models/CLAPWrapper.py:21 plus the injection
    points: `state_dict=` instead of a checkpoint path, `tokenizer=`, `ctx=` to share a backend context)."""

    def __init__(self, model_fp=None, config_path=None, use_cuda=True, state_dict=None, tokenizer=None, ctx=None,
                 device="cuda:0", precision=None, seed=21, crop_start=None, synthetic=None):
        """synthetic=True: seeded random weights of the CLAP architecture (benchmarks and tests -- its scores mean nothing).
        Without it a wrapper built with neither `model_fp` nor `state_dict` still gets them, but says so loudly: the reference
        always loads CLAP_weights_2022.pth (audio-chatgpt.py:146)."""
        self.args = dict(C.CLAP_SCORER)
        self.ctx = ctx or Context(device, precision=precision or default_precision())
        self.device = self.ctx.device
        self.use_cuda = use_cuda
        self.tokenizer = tokenizer
        self.crop_start = crop_start
        if state_dict is None and model_fp is not None:
            state_dict = torch.load(model_fp, map_location="cpu")["model"]                    # (:62)
        if state_dict is None:                                                                # no checkpoint ships: seeded weights
            if not synthetic:
                import warnings
                warnings.warn("CLAPWrapper: no checkpoint (model_fp / state_dict) given -- using SEEDED RANDOM weights; the "
                              "best-of-n ranking they produce is meaningless.  Pass synthetic=True to silence this.", stacklevel=2)
            state_dict = {"caption_encoder." + k: v for k, v in WT.make_clap_text_state_dict(self.args["text"], seed).items()}
            state_dict.update({"audio_encoder." + k: v
                               for k, v in WT.make_clap_audio_state_dict(self.args["audio"], seed + 1).items()})
            state_dict["logit_scale"] = torch.tensor(math.log(1 / 0.07))
        text_sd = {k: v for k, v in _strip(state_dict, "caption_encoder.").items()
                   if not k.startswith("base.pooler.") and not k.endswith("position_ids")}
        audio_sd = _strip(state_dict, "audio_encoder.")
        self.logit_scale = float(state_dict["logit_scale"]) if "logit_scale" in state_dict else math.log(1 / 0.07)
        self.caption_encoder = Encoder(self.ctx, self.args["text"], text_sd)
        self.audio_encoder = ClapAudio(self.ctx, self.args["audio"],
                                       {k: v for k, v in audio_sd.items() if "extractor" not in k and "fc_audioset" not in k})
        a = self.args
        n_fft = a["window_size"]
        rk, ik, mk = ("base.spectrogram_extractor.stft.conv_real.weight", "base.spectrogram_extractor.stft.conv_imag.weight",
                      "base.logmel_extractor.melW")
        if rk in audio_sd and ik in audio_sd and mk in audio_sd:      # the checkpoint's own frozen tables
            basis = torch.cat([audio_sd[rk].reshape(-1, n_fft), audio_sd[ik].reshape(-1, n_fft)], 0).float().numpy()
            melw = audio_sd[mk].float().numpy().T
        else:
            basis = dft_basis(n_fft)
            melw = mel_filterbank(sr=a["sampling_rate"], n_fft=n_fft, n_mels=a["mel_bins"], fmin=a["fmin"], fmax=a["fmax"])
        self.logmel = Spectral(self.ctx, dict(n_fft=n_fft, hop=a["hop_size"], n_mels=a["mel_bins"], pad_mode="reflect", power=2,
                                              log_kind="db", amin=a["amin"], ref=a["ref"], out_layout="btm"), basis, melw)
        self._resamplers = {}

    # ---- audio ------------------------------------------------------------------------------------------------
    def _resampler(self, sample_rate):
        r = self._resamplers.get(sample_rate)
        if r is None:
            k, width = sinc_resample_kernel(sample_rate, self.args["sampling_rate"], **self.args["resample"])
            g = math.gcd(int(sample_rate), self.args["sampling_rate"])
            r = self._resamplers[sample_rate] = Resampler(self.ctx, sample_rate // g, self.args["sampling_rate"] // g, width, k)
        return r

    def resample_and_duration(self, wav_sr, audio_duration, resample=False):
        """CLAPWrapper.resample_and_duration (:103-128) on the device -> 1-D tensor of audio_duration * sample_rate samples
        (sample_rate = the INPUT rate, as upstream)."""
        audio_time_series, sample_rate = wav_sr
        # CLAPWrapper.py:103-110: T.Resample runs on the [channels, n] tensor, THEN everything is flattened (a stereo file's
        # channels are filtered one by one and concatenated; mono -- what T2A produces -- is unaffected)
        x = torch.as_tensor(audio_time_series, dtype=torch.float32).to(self.device)
        x = x.reshape(1, -1) if x.dim() < 2 else x.reshape(-1, x.shape[-1])
        if resample and int(sample_rate) != self.args["sampling_rate"]:
            x = self._resampler(int(sample_rate)).forward(x.contiguous())
        x = x.reshape(-1)
        want = int(audio_duration * sample_rate)
        if want >= x.shape[0]:
            x = x.repeat(int(np.ceil(want / x.shape[0])))[0:want]
        else:
            start = self.crop_start if self.crop_start is not None else random.randrange(x.shape[0] - want)
            x = x[start:start + want]
        return x.contiguous()

    def load_audio_into_tensor(self, audio_path, audio_duration, resample=False):
        try:
            import soundfile
            wav, sr = soundfile.read(audio_path, dtype="float32", always_2d=True)
            wav = wav.T
        except ImportError:
            from scipy.io import wavfile
            sr, wav = wavfile.read(audio_path)
            wav = np.asarray(wav)
            wav = (wav.astype(np.float32) / 32768.0 if wav.dtype == np.int16 else wav.astype(np.float32))
            wav = wav.T if wav.ndim == 2 else wav[None]
        return self.resample_and_duration((wav, sr), audio_duration, resample)

    def preprocess_audio(self, audio_files, resample):
        tensors = []
        for f in audio_files:
            if isinstance(f, str):
                t = self.load_audio_into_tensor(f, self.args["duration"], resample)
            elif isinstance(f, tuple):
                t = self.resample_and_duration(f, self.args["duration"], resample)
            else:
                raise TypeError(f"type of audiofile is {type(f)},which is not supported")
            tensors.append(t.reshape(1, -1))
        return torch.stack(tensors, 0)                                           # default_collate: [N, 1, n]

    def _get_audio_embeddings(self, preprocessed_audio):
        x = preprocessed_audio.reshape(preprocessed_audio.shape[0], preprocessed_audio.shape[2])
        logmel = self.logmel.forward(x)                                          # [N, frames, mel_bins]
        return self.audio_encoder.embed(logmel[:, None])                         # unit length (normalising twice is idempotent)

    def get_audio_embeddings(self, audio_files, resample):
        return self._get_audio_embeddings(self.preprocess_audio(audio_files, resample))

    # ---- text -------------------------------------------------------------------------------------------------
    def preprocess_text(self, text_queries):
        """-> list of 1-D LongTensors: the real (unmasked) tokens of each query, [CLS] first."""
        if self.tokenizer is None:
            raise MaaError("CLAPWrapper.get_text_embeddings needs a tokenizer (AutoTokenizer.from_pretrained("
                           "'bert-base-uncased') with its vocabulary on disk); get_text_embeddings_from_ids takes token ids")
        out = []
        for t in text_queries:
            if hasattr(self.tokenizer, "encode_plus"):
                tok = self.tokenizer.encode_plus(text=t, add_special_tokens=True, max_length=self.args["text_len"],
                                                 padding="max_length", return_tensors="pt")              # (:156-157)
                ids, mask = tok["input_ids"].reshape(-1), tok["attention_mask"].reshape(-1)
                out.append(ids[mask.bool()])
            else:
                out.append(torch.as_tensor(self.tokenizer(t), dtype=torch.long).reshape(-1))
        return out

    def get_text_embeddings_from_ids(self, id_rows):
        """id_rows: list of 1-D id sequences WITHOUT padding -> [N, d_proj], unit length."""
        return torch.cat([self.caption_encoder.encode_cls(torch.as_tensor(r).reshape(1, -1)) for r in id_rows], 0)

    def get_text_embeddings(self, class_labels):
        return self.get_text_embeddings_from_ids(self.preprocess_text(class_labels))

    # ---- scores -----------------------------------------------------------------------------------------------
    def compute_similarity(self, audio_embeddings, text_embeddings, use_logit_scale=True):
        scale = math.exp(self.logit_scale) if use_logit_scale else 1.0
        return clap_similarity(self.ctx, audio_embeddings, text_embeddings, scale)     # [n_audio, n_text] (= similarity.T upstream)

    def cal_clap_score(self, txt, audio_path):
        text_embeddings = self.get_text_embeddings([txt])
        audio_embeddings = self.get_audio_embeddings([audio_path], resample=True)
        return self.compute_similarity(audio_embeddings, text_embeddings, use_logit_scale=False).squeeze().cpu().numpy()
