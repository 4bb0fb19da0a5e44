"""AudioGPT tool classes T2A / I2A / Inpaint with the reference's Python call signatures
(audio-chatgpt.py:140-212, 214-273, 418-558), re-pointed at the MI355X backend.

The LangChain agent binds `Tool(func=self.t2a.inference)` (audio-chatgpt.py:1084), `Tool(func=self.i2a.inference)` (:1114)
and, for inpainting, `Tool(func=self.inpaint.show_mel_fn)` (:1120: audio path in, the path of a viridis PNG of its mel out);
the Gradio callback then calls `Inpaint.inference(input_audio, mel_and_mask)` (:537).  Those signatures are kept
byte-compatible.  The method bodies ARE the reference's glue, kept line for line on purpose (they define the RNG order, the
argument quirks and the file formats the agent sees -- including the `inapint_wav` spelling); what is new is everything
they call: the sampler / model / vocoder / mel objects are the HIP-backed ones.  Reference quirks are reproduced deliberately
(SURVEY.md section 0.8): `inference()` ignores its seed/scale/ddim_steps/n_samples arguments and calls
`txt2audio` with the defaults; `Inpaint.inpaint` builds a seeded start_code and does not pass it.

Pluggable:
  * the conditioning encoders (`model.cond_stage_model`): `cond_stage_model=` takes any object with the reference's
    `encode` / `forward_img` / `preprocess`; when the checkpoint carries `cond_stage_model.*` weights the device towers of
    ldm/encoders.py are built from them (`tokenizer=` / `preprocess=` supply the host-side halves); otherwise a seeded
    stand-in with the encoders' output statistics (ldm/latent_diffusion.SyntheticEmbedder)
  * CLAP best-of-n re-ranking (`select_best_audio`, audio-chatgpt.py:185-199): `clap=` takes a CLAPWrapper
    (audiogpt_amd/clap.py: resampler, log-mel, Cnn14, BERT [CLS] and the similarity all on the device) or the CLAP
    checkpoint's state_dict (then `clap_tokenizer=` supplies the host-side tokenizer); a checkpoint dict that carries
    `clap_model.*` entries builds it too; `scorer=` overrides with any callable (prompt, wav, sr) -> score.  With none of
    them the first sample is returned (the reference always loads CLAP_weights_2022.pth)
  * wav files are written with soundfile when it is installed, scipy.io.wavfile otherwise; PNGs with PIL, viridis from matplotlib
"""
import os
import uuid

import numpy as np
import torch

from . import config as C
from .ldm.ddim import DDIMSampler
from .ldm.latent_diffusion import LatentDiffusionAudio
from .vocoder.hifigan import VocoderBigVGAN

SAMPLE_RATE = 16000


def _write_wav(path, wav, sr):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    try:
        import soundfile
        soundfile.write(path, wav, samplerate=sr)
    except ImportError:
        from scipy.io import wavfile
        wavfile.write(path, sr, np.asarray(wav))


class T2A:
    def __init__(self, device, ckpt_state_dict=None, vocoder_dir=None, cond_stage_model=None, scorer=None,
                 precision=None, tokenizer=None, clap=None, clap_tokenizer=None):
        print("Initializing Make-An-Audio to %s" % device)
        self.device = device
        self.sampler = self._initialize_model(C.LDM_T2A, ckpt_state_dict, device, cond_stage_model, precision, tokenizer)
        self.vocoder = VocoderBigVGAN(vocoder_dir, device=device, ctx=self.sampler.model.ctx)
        self.scorer = scorer
        if clap is None and ckpt_state_dict is not None and any(k.startswith("clap_model.") for k in ckpt_state_dict):
            clap = {k[len("clap_model."):]: v for k, v in ckpt_state_dict.items() if k.startswith("clap_model.")}
        if isinstance(clap, dict):
            from .clap import CLAPWrapper
            clap = CLAPWrapper(state_dict=clap, tokenizer=clap_tokenizer or tokenizer, ctx=self.sampler.model.ctx)
        self.clap_model = clap

    def _initialize_model(self, config, ckpt, device, cond_stage_model=None, precision=None, tokenizer=None):
        model = LatentDiffusionAudio(config, device=device, state_dict=ckpt, cond_stage_model=cond_stage_model,
                                     precision=precision, tokenizer=tokenizer)
        return DDIMSampler(model)

    def txt2audio(self, text, seed=55, scale=1.5, ddim_steps=100, n_samples=3, W=624, H=80):
        prng = np.random.RandomState(seed)
        start_code = prng.randn(n_samples, self.sampler.model.first_stage_model.embed_dim, H // 8, W // 8)
        start_code = torch.from_numpy(start_code).to(device=self.device, dtype=torch.float32)
        uc = self.sampler.model.get_learned_conditioning(n_samples * [""])
        c = self.sampler.model.get_learned_conditioning(n_samples * [text])
        shape = [self.sampler.model.first_stage_model.embed_dim, H // 8, W // 8]
        samples_ddim, _ = self.sampler.sample(S=ddim_steps, conditioning=c, batch_size=n_samples, shape=shape,
                                              verbose=False, unconditional_guidance_scale=scale,
                                              unconditional_conditioning=uc, x_T=start_code)
        x_samples_ddim = self.sampler.model.decode_first_stage(samples_ddim)
        x_samples_ddim = torch.clamp((x_samples_ddim + 1.0) / 2.0, min=0.0, max=1.0)
        wav_list = []
        for idx, spec in enumerate(x_samples_ddim):
            wav = self.vocoder.vocode(spec)
            wav_list.append((SAMPLE_RATE, wav))
        return self.select_best_audio(text, wav_list)

    def select_best_audio(self, prompt, wav_list):
        if self.scorer is not None:
            scores = [float(self.scorer(prompt, wav, sr)) for sr, wav in wav_list]
            return wav_list[int(np.argmax(scores))]
        if self.clap_model is None:
            if not getattr(T2A, "_warned_no_scorer", False):
                T2A._warned_no_scorer = True
                import warnings
                warnings.warn("T2A.select_best_audio: no CLAP scorer configured (clap= / scorer= / clap_model.* in the checkpoint): "
                              "returning the first of the n samples instead of the best one (the reference always loads "
                              "CLAP_weights_2022.pth, audio-chatgpt.py:146)")
            return wav_list[0]
        clap_model = self.clap_model                                                  # audio-chatgpt.py:186-199
        text_embeddings = clap_model.get_text_embeddings([prompt])
        score_list = []
        for data in wav_list:
            sr, wav = data
            audio_embeddings = clap_model.get_audio_embeddings([(torch.FloatTensor(wav), sr)], resample=True)
            score = clap_model.compute_similarity(audio_embeddings, text_embeddings,
                                                  use_logit_scale=False).squeeze().cpu().numpy()
            score_list.append(score)
        max_index = np.array(score_list).argmax()
        print(score_list, max_index)
        return wav_list[max_index]

    def inference(self, text, seed=55, scale=1.5, ddim_steps=100, n_samples=3, W=624, H=80):
        melbins, mel_len = 80, 624
        with torch.no_grad():
            result = self.txt2audio(text=text, H=melbins, W=mel_len)
        audio_filename = os.path.join("audio", str(uuid.uuid4())[0:8] + ".wav")
        _write_wav(audio_filename, result[1], 16000)
        print(f"Processed T2I.run, text: {text}, audio_filename: {audio_filename}")
        return audio_filename


class I2A:
    def __init__(self, device, ckpt_state_dict=None, vocoder_dir=None, cond_stage_model=None, precision=None,
                 preprocess=None):
        print("Initializing Make-An-Audio-Image to %s" % device)
        self.device = device
        model = LatentDiffusionAudio(C.LDM_I2A, device=device, state_dict=ckpt_state_dict,
                                     cond_stage_model=cond_stage_model, seeds=(4, 1), precision=precision,
                                     preprocess=preprocess)
        self.sampler = DDIMSampler(model)
        self.vocoder = VocoderBigVGAN(vocoder_dir, device=device, ctx=model.ctx)

    def img2audio(self, image, seed=55, scale=3, ddim_steps=100, W=624, H=80):
        n_samples = 1
        prng = np.random.RandomState(seed)
        start_code = prng.randn(n_samples, self.sampler.model.first_stage_model.embed_dim, H // 8, W // 8)
        start_code = torch.from_numpy(start_code).to(device=self.device, dtype=torch.float32)
        uc = self.sampler.model.get_learned_conditioning(n_samples * [""])
        if isinstance(image, str):
            from PIL import Image
            image = Image.open(image)
        image = self.sampler.model.cond_stage_model.preprocess(image).unsqueeze(0)
        image_embedding = self.sampler.model.cond_stage_model.forward_img(image)
        c = image_embedding.repeat(n_samples, 1, 1).to(self.device)
        shape = [self.sampler.model.first_stage_model.embed_dim, H // 8, W // 8]
        samples_ddim, _ = self.sampler.sample(S=ddim_steps, conditioning=c, batch_size=n_samples, shape=shape,
                                              verbose=False, unconditional_guidance_scale=scale,
                                              unconditional_conditioning=uc, x_T=start_code)
        x_samples_ddim = self.sampler.model.decode_first_stage(samples_ddim)
        x_samples_ddim = torch.clamp((x_samples_ddim + 1.0) / 2.0, min=0.0, max=1.0)
        wav_list = []
        for idx, spec in enumerate(x_samples_ddim):
            wav_list.append((SAMPLE_RATE, self.vocoder.vocode(spec)))
        return wav_list[0]

    def inference(self, image, seed=55, scale=3, ddim_steps=100, W=624, H=80):
        melbins, mel_len = 80, 624
        with torch.no_grad():
            result = self.img2audio(image=image, H=melbins, W=mel_len)
        audio_filename = os.path.join("audio", str(uuid.uuid4())[0:8] + ".wav")
        _write_wav(audio_filename, result[1], 16000)
        print(f"Processed I2a.run, image_filename: {image}, audio_filename: {audio_filename}")
        return audio_filename


class Inpaint:
    def __init__(self, device, ckpt_state_dict=None, vocoder_dir=None, mel_transform=None, precision=None):
        print("Initializing Make-An-Audio-inpaint to %s" % device)
        self.device = device
        model = LatentDiffusionAudio(C.LDM_INPAINT, device=device, state_dict=ckpt_state_dict, seeds=(5, 1),
                                     precision=precision)
        self.sampler = DDIMSampler(model)
        self.vocoder = VocoderBigVGAN(vocoder_dir, device=device, ctx=model.ctx)
        # gen_mel_audio + TRANSFORMS_16000 (audio-chatgpt.py:468-491, extract_mel_spectrogram.py:140-150) on the device
        # (STFT and filter bank as two GEMMs, audiogpt_amd/mel.DeviceMelTransform) unless the caller plugs in the
        # librosa-backed original
        if mel_transform is None:
            from .mel import DeviceMelTransform
            mel_transform = DeviceMelTransform(model.ctx)
        self.mel_transform = mel_transform
        import matplotlib.cm
        self.cmap_transform = matplotlib.cm.viridis                                   # audio-chatgpt.py:424

    def make_batch_sd(self, mel, mask, num_samples=1):
        mel = torch.from_numpy(mel)[None, None, ...].to(dtype=torch.float32)
        mask = torch.from_numpy(mask)[None, None, ...].to(dtype=torch.float32)
        masked_mel = (1 - mask) * mel
        mel = mel * 2 - 1
        mask = mask * 2 - 1
        masked_mel = masked_mel * 2 - 1
        rep = lambda t: t.to(device=self.device).repeat(num_samples, 1, 1, 1)   # noqa: E731
        return {"mel": rep(mel), "mask": rep(mask), "masked_mel": rep(masked_mel)}

    def gen_mel(self, input_audio_path):
        """audio-chatgpt.py:452-467: wav file -> [80, frames] mel in [0, 1] (int16 -> float, stereo -> mono, librosa.resample
        to 16 kHz, crop / zero-extend to the clip length, TRANSFORMS_16000 -- all inside `self.mel_transform`)."""
        from scipy.io import wavfile
        sr, ori_wav = wavfile.read(input_audio_path)
        return self.mel_transform(sr, ori_wav)

    def gen_mel_audio(self, input_audio):
        """audio-chatgpt.py:468-491: the same for the `(sr, samples)` pair Gradio hands over."""
        sr, ori_wav = input_audio
        return self.mel_transform(sr, ori_wav)

    def show_mel_fn(self, input_audio_path):
        """The "Audio Inpainting" Tool.func (audio-chatgpt.py:492-499, registered at :1120): audio path -> 'image/<8 hex>.png',
        the first 500 mel frames through viridis."""
        from PIL import Image
        crop_len = 500
        crop_mel = self.gen_mel(input_audio_path)[:, :crop_len]
        color_mel = self.cmap_transform(crop_mel)
        image = Image.fromarray((color_mel * 255).astype(np.uint8))
        image_filename = os.path.join("image", str(uuid.uuid4())[0:8] + ".png")
        os.makedirs("image", exist_ok=True)
        image.save(image_filename)
        return image_filename

    def inpaint(self, batch, seed, ddim_steps, num_samples=1, W=512, H=512):
        model = self.sampler.model
        prng = np.random.RandomState(seed)
        start_code = prng.randn(num_samples, model.first_stage_model.embed_dim, H // 8, W // 8)
        start_code = torch.from_numpy(start_code).to(device=self.device, dtype=torch.float32)   # unused, as upstream
        c = model.get_first_stage_encoding(model.encode_first_stage(batch["masked_mel"]))
        cc = torch.nn.functional.interpolate(batch["mask"], size=c.shape[-2:])
        c = torch.cat((c, cc), dim=1)
        shape = (c.shape[1] - 1,) + c.shape[2:]
        samples_ddim, _ = self.sampler.sample(S=ddim_steps, conditioning=c, batch_size=c.shape[0], shape=shape,
                                              verbose=False)
        x_samples_ddim = model.decode_first_stage(samples_ddim)
        mel = torch.clamp((batch["mel"] + 1.0) / 2.0, min=0.0, max=1.0)
        mask = torch.clamp((batch["mask"] + 1.0) / 2.0, min=0.0, max=1.0)
        predicted_mel = torch.clamp((x_samples_ddim + 1.0) / 2.0, min=0.0, max=1.0)
        inpainted = (1 - mask) * mel + mask * predicted_mel
        inpainted = inpainted.cpu().numpy().squeeze()
        inapint_wav = self.vocoder.vocode(inpainted)
        return inpainted, inapint_wav

    def inference_mel(self, input_mel, mask, seed=55, ddim_steps=100):
        """The device part of `inference` (audio-chatgpt.py:539-548) for an [80, 848] mel in [0,1] and a mask."""
        mel_bins, mel_len = 80, 848
        input_mel = input_mel[:, :mel_len]
        mask = np.pad(mask, ((0, 0), (0, mel_len - mask.shape[1])), mode="constant", constant_values=0)
        with torch.no_grad():
            batch = self.make_batch_sd(input_mel.astype(np.float32), mask.astype(np.float32), num_samples=1)
            return self.inpaint(batch=batch, seed=seed, ddim_steps=ddim_steps, num_samples=1, H=mel_bins, W=mel_len)

    def inference(self, input_audio, mel_and_mask, seed=55, ddim_steps=100):
        from PIL import Image
        torch.set_grad_enabled(False)
        show_mel = np.array(Image.open(mel_and_mask["image"]).convert("L")) / 255
        mask = np.array(Image.open(mel_and_mask["mask"]).convert("L")) / 255
        input_mel = self.gen_mel_audio(input_audio)
        inpainted, gen_wav = self.inference_mel(input_mel, mask, seed, ddim_steps)
        inpainted = inpainted[:, :show_mel.shape[1]]
        color_mel = self.cmap_transform(inpainted)
        input_len = int(input_audio[1].shape[0] * SAMPLE_RATE / input_audio[0])
        gen_wav = (gen_wav * 32768).astype(np.int16)[:input_len]
        image = Image.fromarray((color_mel * 255).astype(np.uint8))
        image_filename = os.path.join("image", str(uuid.uuid4())[0:8] + ".png")
        os.makedirs("image", exist_ok=True)
        image.save(image_filename)
        audio_filename = os.path.join("audio", str(uuid.uuid4())[0:8] + ".wav")
        _write_wav(audio_filename, gen_wav, 16000)
        return image_filename, audio_filename
