"""The secondary workloads of the default line: BASELINE configs[2] (hifigan64), configs[4] (mixed), configs[1] in plain bf16 and
with BigVGAN, and the latency of one tool call."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audiogpt_amd import config as C            # noqa: E402
from audiogpt_amd import weights as WT          # noqa: E402
from .common import (CFG_SCALE, CLIP_FRAMES, DDIM_STEPS, HIFIGAN64, LATENT, MFMA_PER_FLOP, PEAK_TFLOPS, PROMPTS_PER_GPU,      # noqa: E402
                     _t2a_inputs, _timed, hifigan64_mel, synth_conditioning)
from .cpu import cpu_baseline, cpu_baseline_mixed      # noqa: E402
from .roofline import attach_traffic, roofline_of      # noqa: E402


def run_hifigan64(dev, precision, steps, warmup, cpu_base=True, roofline=True):
    """BASELINE configs[2] on one GPU.  A step = one generator pass over the [64, 80, 1024] mel batch resident in HBM."""
    from audiogpt_amd.backend import Context, Vocoder
    cfg = C.HIFIGAN_NS_512
    ctx = Context(dev, precision=precision)
    voc = Vocoder(ctx, cfg, WT.make_vocoder_state_dict(cfg, seed=2))
    mel = hifigan64_mel().to(dev)
    B, T = mel.shape[0], mel.shape[2]
    for _ in range(warmup):
        voc(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wav = voc(mel)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert wav.shape[-1] == T * voc.hop
    audio_s = B * T * voc.hop / float(cfg["sampling_rate"])
    res = {"metric": "vocoded audio-seconds/sec (HiFi-GAN 22.05 kHz, 64 x 1024 frames)", "value": audio_s * steps / elapsed,
           "unit": "audio-seconds/sec", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "higher_is_better": True, "dtype": precision, "data": "synthetic mel clip(N(-2.25,1.5),-6,1.5) seed 7; seeded random-init weights",
           "config": {"workload": "NeuralSeq HiFi-GAN generator only, upsample_initial_channel 512, batch 64 x 1024 frames -> 64 x 262144 samples",
                      "audio_seconds_per_step": audio_s}}
    if roofline:
        ctx.prof_begin()
        voc(mel)
        rows = ctx.prof_end()
        r = roofline_of(rows, precision)
        # the pass as a whole: SURVEY 8(d) prices it at 0.6288 TFLOP per 1024-frame item and, in fp32 storage, at
        # 218 MB per item of stage-boundary bytes (the fused ideal) / 4.15 GB per item layer by layer
        total_ms = sum(v["ms"] for v in rows.values())
        r["whole_pass"] = {"tflop": 0.6288 * B, "achieved_tflops": 0.6288 * B / (total_ms * 1e-3),
                           "frac_of_mfma_peak": 0.6288 * B / (total_ms * 1e-3) / r["peak"],
                           "kernel_ms": total_ms,
                           "hbm_gbs_if_layer_by_layer": 4.15 * B / (total_ms * 1e-3), "hbm_gbs_if_fused_ideal": 0.218 * B / (total_ms * 1e-3),
                           "hbm_peak_gbs": 8000.0}
        attach_traffic(r, precision, "hifigan64", 1)
        res["roofline"] = r
    if cpu_base:
        from oracle import vocoder as O_voc
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        gsd = O_voc.fold_weight_norm(WT.make_vocoder_state_dict(cfg, seed=2))
        m1 = mel[:1].cpu()
        with torch.no_grad():
            O_voc.hifigan_forward(gsd, cfg, m1[:, :, :64])
            t0 = time.perf_counter()
            n_items = 0
            while time.perf_counter() - t0 < 10.0:
                O_voc.hifigan_forward(gsd, cfg, m1)
                n_items += 1
            dt = time.perf_counter() - t0
        res["cpu_baseline"] = dict(value=n_items * T * voc.hop / float(cfg["sampling_rate"]) / dt, unit="audio-seconds/sec",
                                   cores=cores, kind="port",
                                   sample="%d item(s) of 1024 frames through the CPU oracle (%.2f s each); torch %s fp32, %d threads"
                                          % (n_items, dt / n_items, torch.__version__, cores))
    voc.close()
    ctx.close()
    return res


def mixed_inputs(n=PROMPTS_PER_GPU):
    """Synthetic inputs of the mixed tool batch (CPU tensors; tests/golden/make_golden.py `mixed` replays single rows of them
    through the reference): mels U(0,1) [n,1,80,848] with rectangle masks, L2-normalised N(0,1) image embeddings
    [n,1,1024], one layer-normed unconditional row, posterior noise and start codes."""
    g = torch.Generator().manual_seed(77)
    mel = torch.rand(n, 1, 80, 848, generator=g)
    mask = torch.zeros(n, 1, 80, 848)
    for b in range(n):
        t0, f0 = 100 + 40 * b, 8 + 3 * b
        mask[b, :, f0:f0 + 40, t0:t0 + 300] = 1.0
    emb = torch.randn(n, 1, 1024, generator=g)
    emb = emb / emb.norm(dim=-1, keepdim=True)
    uc = torch.nn.functional.layer_norm(torch.randn(1, 1, 1024, generator=g), (1024,)).expand(n, -1, -1).contiguous()
    noise = torch.randn(n, 4, 10, 106, generator=g)
    xT_inp = torch.randn(n, 4, 10, 106, generator=g)
    xT_i2a = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float()
    return mel, mask, emb, uc, noise, xT_inp, xT_i2a


def mixed_inpaint(inp, mel, mask, noise, xT, S, use_graph=True):
    """tools.Inpaint.inpaint, batched, on pipeline `inp` -> (waveforms, composited mels, latents)."""
    mom = inp.vae.encode_moments((1 - mask) * mel * 2 - 1)
    mean, logvar = mom.chunk(2, dim=1)
    zc = mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise
    cc = torch.nn.functional.interpolate(mask * 2 - 1, size=zc.shape[-2:])
    z = inp.sample_latents(xT, S=S, concat=torch.cat((zc, cc), dim=1), use_graph=use_graph)
    pred = inp.decode(z)[:, None]
    comp = (1 - mask) * mel + mask * pred
    return inp.vocode(comp[:, 0]), comp[:, 0], z


def run_mixed(dev, precision, steps, warmup, n=PROMPTS_PER_GPU, S=DDIM_STEPS, roofline=True, cpu_base=True):
    """BASELINE configs[4] on one GPU: a mixed tool batch, each tool's DDIM step captured as a hipGraph.
      inpaint: n masked mels [80, 848] (U(0,1), random rectangle masks) -> VAE encode + posterior sample -> concat-conditioned
               DDIM over [n, 9, 10, 106] without CFG (inpaint beta schedule) -> decode -> composite with the input mel ->
               BigVGAN (848 frames, 13.568 s each)                                       (audio-chatgpt.py:500-528)
      i2a:     n image embeddings (L2-normalised N(0,1) [n, 1, 1024]) -> DDIM with CFG 3 over a 1-token context (UNet batch
               2n, context also added to the time embedding) -> decode -> BigVGAN (624 frames, 9.984 s each)   (:232-261)
    A step = both tools once; value = audio-seconds of both per wall second."""
    from audiogpt_amd.pipeline import MakeAnAudio
    inp = MakeAnAudio(dev, ldm=C.LDM_INPAINT, vocoder_cfg=C.BIGVGAN_16K, seeds=(5, 1, 3), with_encoder=True, precision=precision)
    i2a = MakeAnAudio(dev, ldm=C.LDM_I2A, vocoder_cfg=C.BIGVGAN_16K, seeds=(4, 1, 3), precision=precision)
    mel, mask, emb, uc, noise, xT_inp, xT_i2a = (t.to(dev) for t in mixed_inputs(n))

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1, initializer=torch.cuda.set_device, initargs=(dev,))

    def one_step():
        # the two tools are independent requests on their own pipelines (streams): image -> audio runs beside inpainting
        f2 = pool.submit(lambda: i2a.generate(xT_i2a, emb, uc, 3.0, S)[0])
        w1 = mixed_inpaint(inp, mel, mask, noise, xT_inp, S)[0]
        return w1, f2.result()

    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w1, w2 = one_step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    audio_s = (w1.shape[0] * w1.shape[1] + w2.shape[0] * w2.shape[1]) / 16000.0
    res = {"metric": "generated audio-seconds/sec (mixed tool batch: inpaint 13.6 s clips + image-to-audio 10 s clips, 100 DDIM steps)",
           "value": audio_s * steps / elapsed, "unit": "audio-seconds/sec", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "dtype": precision,
           "data": "synthetic mels U(0,1) with rectangle masks, L2-normalised N(0,1) image embeddings; seeded random-init weights",
           "config": {"workload": "inpaint x%d ([%d,9,10,106], no CFG) + image-to-audio x%d (CFG 3, 1-token context), %d DDIM steps each, "
                                  "VAE + BigVGAN, hipGraph-captured steps" % (n, n, n, S), "audio_seconds_per_step": audio_s}}
    if roofline:
        inp.ctx.prof_begin()
        i2a.ctx.prof_begin()
        mixed_inpaint(inp, mel, mask, noise, xT_inp, S, use_graph=False)
        i2a.generate(xT_i2a, emb, uc, 3.0, S, use_graph=False)
        rows = inp.ctx.prof_end()
        for k, v in i2a.ctx.prof_end().items():
            if k in rows:
                for f in ("launches", "ms", "flops", "bytes"):
                    rows[k][f] += v[f]
            else:
                rows[k] = v
        res["roofline"] = roofline_of(rows, precision)
        attach_traffic(res["roofline"], precision, "mixed", S)
    inp.close()
    i2a.close()
    if cpu_base:
        res["cpu_baseline"] = cpu_baseline_mixed(S=S)
    return res

def run_t2a_variant(dev, precision, vocoder_cfg, label, cpu_parts=None, steps=6, inflight=3, parity_against=None,
                    roofline=True, cpu_base=True):
    """BASELINE configs[1] (8 prompts x 100 CFG DDIM steps -> VAE -> vocoder) under another precision mode or vocoder, measured
    like the headline: `inflight` batches of 8 in flight for `value`, one batch alone for `one_batch_in_flight`.
      parity_against = a precision mode: the same batch in that mode (bf16x3 meets the fp32 gates at mel-L1 3.5e-6)
                       -> mel-L1 on the [0,1] mel and waveform RMS between the two
      parity_against = "oracle_vocoder": the CPU oracle's vocoder on one of the produced mels -> waveform RMS (the stage that
                       differs from the headline)."""
    from concurrent.futures import ThreadPoolExecutor

    from audiogpt_amd.pipeline import MakeAnAudio
    n, S = PROMPTS_PER_GPU, DDIM_STEPS
    pipes = [MakeAnAudio(dev, vocoder_cfg=vocoder_cfg, precision=precision, stream=torch.cuda.Stream(dev)) for _ in range(inflight)]
    for p_ in pipes:
        p_.ctx.set_concurrency(inflight)
        p_.ctx.set_cfg_split(inflight == 1)      # (as the headline: lanes only when one batch owns the GPU)
    x_T, c, uc = _t2a_inputs(n, dev)
    pool = ThreadPoolExecutor(max_workers=inflight, initializer=torch.cuda.set_device, initargs=(dev,))
    gen = lambda p_: p_.generate(x_T, c, uc, CFG_SCALE, S)      # noqa: E731

    def round_of(k):          # k batches, `inflight` at a time
        futs = [pool.submit(gen, pipes[i % inflight]) for i in range(k)]
        return [f.result() for f in futs][-1]
    round_of(inflight)        # warm-up: every replica sizes its workspace and captures its step graph
    per_step, _ = _timed(lambda: round_of(steps), 1)
    per_step /= steps
    pipes[0].ctx.set_concurrency(1)
    pipes[0].ctx.set_cfg_split(True)
    gen(pipes[0])                                  # (the step graph of the two-lane form)
    one, (wav, spec, z) = _timed(lambda: gen(pipes[0]), 2)
    pipes[0].ctx.set_concurrency(inflight)
    pipes[0].ctx.set_cfg_split(inflight == 1)
    audio_s = pipes[0].audio_seconds(n, CLIP_FRAMES)
    res = {"metric": "generated audio-seconds/sec (10s clip, 100 DDIM steps) [%d independent batches of %d prompts in flight]" % (inflight, n),
           "value": audio_s / per_step, "unit": "audio-seconds/sec", "n_gpus": 1, "steps": steps, "warmup": 1,
           "ms_per_step": 1e3 * per_step, "higher_is_better": True, "dtype": precision,
           "data": "synthetic prompts (layer-normed N(0,1) [B,77,1024]); seeded random-init weights",
           "config": {"workload": label, "prompts_per_gpu": n, "ddim_steps": S, "batches_in_flight": inflight,
                      "audio_seconds_per_step": audio_s},
           "one_batch_in_flight": {"value": audio_s / one, "ms_per_step": 1e3 * one}}
    if roofline:      # ONE eager stream: the context is told it owns the GPU (tiles by launch time), as in bench.main's roofline pass
        pipes[0].ctx.set_concurrency(1)
        pipes[0].ctx.set_cfg_split(False)
        pipes[0].ctx.prof_begin()
        pipes[0].generate(x_T, c, uc, CFG_SCALE, S, use_graph=False)
        res["roofline"] = roofline_of(pipes[0].ctx.prof_end(), precision)
        pipes[0].ctx.set_concurrency(inflight)
        pipes[0].ctx.set_cfg_split(inflight == 1)
    if parity_against in ("f32", "bf16x3", "bf16"):
        ref = MakeAnAudio(dev, vocoder_cfg=vocoder_cfg, precision=parity_against)
        wav_r, spec_r, _ = ref.generate(x_T, c, uc, CFG_SCALE, S)
        l1 = float((spec - spec_r).abs().mean())
        rms = float(((wav - wav_r) ** 2).mean().sqrt())
        res["parity"] = {"against": "the same batch in the %s mode (itself gated against reference goldens at mel-L1 / wav-RMS <= 1e-4: "
                                    "tests/test_gpu_config2.py)" % parity_against,
                         "mel_l1": l1, "wav_rms": rms, "gate": 1e-4, "meets_gate": bool(l1 <= 1e-4 and rms <= 1e-4)}
        ref.close()
    elif parity_against == "oracle_vocoder":
        from oracle import vocoder as O_voc
        gsd = O_voc.fold_weight_norm(WT.make_vocoder_state_dict(vocoder_cfg, seed=2))
        fwd = O_voc.bigvgan_forward if vocoder_cfg["kind"] == "bigvgan" else O_voc.hifigan_forward
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        t0 = time.perf_counter()
        with torch.no_grad():
            w_ref = fwd(gsd, vocoder_cfg, spec[:1].cpu())
        t_voc = time.perf_counter() - t0
        res["_oracle_vocoder_s"] = t_voc
        rms = float(((wav[0].cpu() - w_ref.reshape(-1)) ** 2).mean().sqrt())
        res["parity"] = {"against": "the CPU oracle's vocoder on the first clip's mel (oracle pinned to the reference generator: "
                                    "tests/test_oracle_golden.py); UNet / VAE parity as the headline's", "wav_rms": rms, "gate": 1e-4,
                         "meets_gate": bool(rms <= 1e-4)}
        if cpu_base and cpu_parts:
            total = DDIM_STEPS * cpu_parts["unet_cfg_step_s"] + cpu_parts["vae_decode_s"] + t_voc
            res["cpu_baseline"] = dict(value=(CLIP_FRAMES * 256 / 16000.0) / total, unit="audio-seconds/sec", cores=min(os.cpu_count() or 1, 32),
                                       kind="port", sample="the headline's CPU-oracle UNet step (%.2f s, scaled x%d) and VAE decode (%.2f s) + this "
                                       "vocoder's oracle pass over one 624-frame clip (%.2f s)" % (cpu_parts["unet_cfg_step_s"], DDIM_STEPS,
                                                                                               cpu_parts["vae_decode_s"], t_voc))
    if "cpu_baseline" not in res and cpu_base and cpu_parts:
        total = DDIM_STEPS * cpu_parts["unet_cfg_step_s"] + cpu_parts["vae_decode_s"] + cpu_parts["hifigan_624_s"]
        res["cpu_baseline"] = dict(value=(CLIP_FRAMES * 256 / 16000.0) / total, unit="audio-seconds/sec", cores=min(os.cpu_count() or 1, 32),
                                   kind="port", sample="the headline's CPU-oracle timings (the same workload in fp32): see cpu_baseline of the line")
    for p_ in pipes:
        p_.close()
    pool.shutdown()
    return res


def run_tool_latency(dev, precision, cpu_parts=None, cpu_base=True, roofline=True):
    """One call of each tool as the reference makes it, from the Python call to the waveform on the host (north_star: real-time or
    better 10-s text -> audio at 100 DDIM steps):
      T2A.txt2audio  n_samples = 3 with CFG 1.5 (UNet batch 6), VAE, BigVGAN once per sample, CLAP best-of-3 on the device
                     (audio-chatgpt.py:158-199)
      I2A.img2audio  n = 1, CFG 3 over a 1-token context, VAE, BigVGAN (audio-chatgpt.py:232-261)
    single stream, hipGraph-captured DDIM steps; seeded random-init weights (CLAP included), synthetic conditioning encoders."""
    import contextlib

    from audiogpt_amd.clap import CLAPWrapper
    from audiogpt_amd.tools import I2A, T2A
    quiet = lambda: contextlib.redirect_stdout(sys.stderr)      # noqa: E731  (the tools print like the reference's; stdout is the JSON line's)

    class Tok:          # the host-side tokenizer is a constructor argument (its vocabulary file does not ship): fixed ids
        def __call__(self, text):
            return [101, 2023, 2003, 1037, 3231, 102]
    out = {"metric": "tool latency, call to waveform (ms)", "unit": "ms", "higher_is_better": False, "dtype": precision, "n_gpus": 1,
           "data": "seeded random-init weights (UNet, VAE, BigVGAN, CLAP); synthetic text / image embeddings",
           "config": {"workload": "T2A.txt2audio(n_samples=3, scale=1.5, ddim_steps=100) + CLAP best-of-3; I2A.img2audio(n=1, scale=3, ddim_steps=100)"}}
    with quiet():
        t2a = T2A(dev, precision=precision)
    t2a.clap_model = CLAPWrapper(ctx=t2a.sampler.model.ctx, tokenizer=Tok(), crop_start=0, synthetic=True)
    text = "a dog barks while rain falls on a tin roof"
    with torch.no_grad(), quiet():
        t2a.txt2audio(text)                                   # first call: workspace + graph capture
        ms_t2a, (sr, wav) = _timed(lambda: t2a.txt2audio(text), 2)
    clip_s = wav.shape[0] / float(sr)
    out["T2A_txt2audio"] = {"ms": 1e3 * ms_t2a, "clip_seconds": clip_s, "realtime_factor": clip_s / ms_t2a,
                            "candidate_audio_seconds_per_sec": 3 * clip_s / ms_t2a}
    if roofline:
        t2a.sampler.model.ctx.prof_begin()
        with torch.no_grad(), quiet():
            t2a.txt2audio(text)
        rows = t2a.sampler.model.ctx.prof_end()
        out["roofline"] = roofline_of(rows, precision)
        out["roofline"]["note"] = "T2A.txt2audio call, graph replay as shipped (kernels inside graph launches are not event-timed: this table covers the eager part -- VAE, BigVGAN, CLAP)"
    img = np.random.RandomState(3).rand(64, 64, 3).astype(np.float32)
    with quiet():
        i2a = I2A(dev, precision=precision)
    with torch.no_grad(), quiet():
        i2a.img2audio(img)
        ms_i2a, (sr2, wav2) = _timed(lambda: i2a.img2audio(img), 2)
    out["I2A_img2audio"] = {"ms": 1e3 * ms_i2a, "clip_seconds": wav2.shape[0] / float(sr2), "realtime_factor": wav2.shape[0] / float(sr2) / ms_i2a}
    out["value"] = 1e3 * ms_t2a
    out["parity"] = {"against": "the same calls are gated in tests/test_gpu_tools.py (T2A.txt2audio / I2A.img2audio vs the CPU oracle chain, "
                                "wav-RMS <= 1e-4) and tests/test_gpu_clap.py (scorer vs the reference's wav_evaluation classes)"}
    if cpu_base and cpu_parts:
        # the same call on the CPU oracle, from the components timed for the other lines: 100 CFG UNet steps at batch 2 x 3 samples,
        # 3 VAE decodes, 3 BigVGAN passes (the scorer is left out: under 1 % of it)
        big = cpu_parts.get("bigvgan_624_s")
        if big is not None:
            total = 3 * (DDIM_STEPS * cpu_parts["unet_cfg_step_s"] + cpu_parts["vae_decode_s"] + big)
            out["cpu_baseline"] = dict(value=1e3 * total, unit="ms", cores=min(os.cpu_count() or 1, 32), kind="port",
                                       sample="3 samples x (100 x %.2f s CFG UNet step + %.2f s VAE decode + %.2f s BigVGAN), the CPU-oracle "
                                              "timings of the other lines of this run" % (cpu_parts["unet_cfg_step_s"], cpu_parts["vae_decode_s"], big))
    t2a.sampler.model.ctx.synchronize()
    return out


def one_batch_records(result, pipe, x_T, c_all, uc_row, n, S, use_graph, lanes, lanes_one, stub, steps, barrier):
    """`one_batch_in_flight`: the same K steps strictly one batch after another -- ONE batch of n prompts owning the GPU, BASELINE
    configs[1] read literally (the latency-oriented number) -- and the same batch with the CFG halves of a step in the other form
    (one stream <-> two lanes), which must give the same waveforms bit for bit."""
    # the same K steps strictly one batch after another on one stream (the latency-oriented number)
    k1 = min(steps, 3)
    if not stub:
        pipe.ctx.set_concurrency(1)      # ONE batch owns the GPU now
        pipe.ctx.set_cfg_split(lanes_one)
        pipe.generate(x_T, c_all[:n], uc_row.expand(n, -1, -1).contiguous(), CFG_SCALE, S, use_graph=use_graph)      # (its step graph)
    barrier()
    t0 = time.perf_counter()
    for _ in range(k1):
        pipe.generate(x_T, c_all[:n], uc_row.expand(n, -1, -1).contiguous(), CFG_SCALE, S, use_graph=use_graph)
    barrier()
    one = time.perf_counter() - t0
    result["one_batch_in_flight"] = {"value": pipe.audio_seconds(n, CLIP_FRAMES) * k1 / one, "ms_per_step": 1e3 * one / k1,
                                     "steps": k1, "cfg_lanes": 2 if lanes_one else 1}
    result["batch_latency_ms"]["alone"] = 1e3 * one / k1
    if not stub:
        # the same batch with the two halves of every CFG step one after the other on ONE stream (the library's default runs
        # them as two lanes -- two branches of the captured step graph, csrc/ddim.cpp): the A/B of that default, and the
        # check that both forms give the same waveforms bit for bit
        c1, uc1 = c_all[:n], uc_row.expand(n, -1, -1).contiguous()
        w_lanes = pipe.generate(x_T, c1, uc1, CFG_SCALE, S, use_graph=use_graph)[0]
        pipe.ctx.set_cfg_split(not lanes_one)
        pipe.generate(x_T, c1, uc1, CFG_SCALE, S, use_graph=use_graph)      # (captures the other form's step graph)
        barrier()
        t0 = time.perf_counter()
        for _ in range(k1):
            w_other = pipe.generate(x_T, c1, uc1, CFG_SCALE, S, use_graph=use_graph)[0]
        barrier()
        two = time.perf_counter() - t0
        pipe.ctx.set_cfg_split(lanes)
        result["one_batch_other_form"] = {
            "value": pipe.audio_seconds(n, CLIP_FRAMES) * k1 / two, "ms_per_step": 1e3 * two / k1, "steps": k1,
            "cfg_lanes": 1 if lanes_one else 2, "bit_identical": bool(torch.equal(w_lanes, w_other)),
            "method": "the same batch with the CFG halves of a step %s" % ("on one stream" if lanes_one else "as two lanes")}
        result["batch_latency_ms"]["alone_other_form"] = 1e3 * two / k1


def run_secondaries(result, pipes, dev, args):
    """The secondary workloads of the default N = 1 line (never lose the headline to one of them)."""
    for p_ in pipes:
        p_.close()
    result["secondary"] = {}
    parts = dict((result.get("cpu_baseline") or {}).get("parts") or {})
    cb, rf = not args.no_cpu_baseline, not args.no_roofline

    def bigvgan_line():
        r = run_t2a_variant(dev, args.precision, C.BIGVGAN_16K, "configs[1] with the vocoder the tool loads (BigVGAN, audio-chatgpt.py:145) "
                            "instead of HiFi-GAN(16k): T2A batch=8, 100 DDIM steps, CFG 1.5, UNet+VAE+BigVGAN", parts or None,
                            parity_against="oracle_vocoder", roofline=rf, cpu_base=cb)
        t_voc = r.pop("_oracle_vocoder_s", None)
        if t_voc is not None:
            parts["bigvgan_624_s"] = t_voc
        return r
    for name, fn in (("hifigan64", lambda: run_hifigan64(dev, args.precision, 3, 1, cb, rf)),
                     ("mixed", lambda: run_mixed(dev, args.precision, 2, 1, roofline=rf, cpu_base=cb)),
                     # BASELINE configs[1] says "bf16": the same workload with operands rounded to bf16 (one MFMA per multiply-add).
                     # Reported, never the headline: it misses the 1e-4 gates (see its parity record)
                     ("t2a_bf16", lambda: run_t2a_variant(dev, "bf16", C.HIFIGAN_16K, "BASELINE configs[1] literally in bf16: T2A batch=8, 100 "
                                                          "DDIM steps, CFG 1.5, UNet+VAE+HiFi-GAN(16k), fp32 storage, bf16 MFMA operands "
                                                          "(EXPECTED TO MISS the 1e-4 mel-L1 / wav-RMS gates: see parity)", parts or None,
                                                          parity_against=args.precision if args.precision != "bf16" else "bf16x3",
                                                          roofline=rf, cpu_base=cb)),
                     ("t2a_bigvgan", bigvgan_line),
                     ("tool_latency", lambda: run_tool_latency(dev, args.precision, parts or None, cb, rf))):
        if args.secondary_only and name not in args.secondary_only.split(","):
            continue
        try:
            result["secondary"][name] = fn()
        except Exception as e:      # never lose the headline line to a secondary workload
            result["secondary"][name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
