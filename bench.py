"""Throughput benchmark of the Make-An-Audio hot path on MI355X.

    python bench.py --gpus 1 --steps 6 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], per GPU): T2A, 8 prompts -> 8 latents [4,10,78], 100 DDIM steps with
classifier-free guidance 1.5 (UNet batch 16), VAE decode to [8,80,624] mels, HiFi-GAN (hifi_0127 args) to
8 x 159744 samples @16 kHz = 79.87 audio-seconds.  One "step" of this benchmark = one such batch through the
whole path.  Synthetic conditioning (layer-normed N(0,1) [B,77,1024], as CLAP's Projection emits) and seeded
random-init weights of the reference architecture (no checkpoints ship with the reference).
With N > 1 (BASELINE configs[3]) each rank owns 8 different prompts (weak scaling): rank 0 produces the
conditioning for all ranks and broadcasts it over RCCL, waveforms are gathered to rank 0; no collective runs
inside the DDIM loop.

Secondary workloads (reported under "secondary" in the same JSON line at N = 1; hifigan64 / mixed also alone with --workload):
  t2a_bf16     -- configs[1] literally in bf16 (one MFMA per multiply-add) with its measured mel-L1 / wav-RMS against the bf16x3 run
  t2a_bigvgan  -- configs[1] with BigVGAN, the vocoder the T2A tool actually loads (audio-chatgpt.py:145)
  tool_latency -- one T2A.txt2audio call (n_samples 3, CFG, BigVGAN per sample, CLAP best-of-3) and one I2A.img2audio call, ms
  hifigan64 -- BASELINE configs[2]: NeuralSeq HiFi-GAN (22.05 kHz, upsample_initial_channel 512), mel [64, 80, 1024]
               (clip(N(-2.25, 1.5), -6, 1.5), seed 7) -> wave [64, 262144] = 760.9 audio-seconds per step, mel resident
               in HBM; roofline of its dominant kernel and of the whole pass against the 40.24 TFLOP it computes.

  mixed     -- BASELINE configs[4] on one GPU: 8 inpaint clips (VAE encode, concat-conditioned DDIM [8,9,10,106] without
               CFG, decode, compositing, BigVGAN 848 frames) + 8 image-to-audio clips (1-token context, CFG 3, BigVGAN 624
               frames), 100 DDIM steps each, every step a hipGraph replay = 188.4 audio-seconds per step.

Steps overlap: `--inflight 3` (default) keeps three consecutive steps -- independent batches of 8 prompts -- in flight per GPU on
three pipeline replicas (streams); every batch is still sampled as configs[1] says.  The METHOD IS PART OF THE METRIC STRING
("... [3 batches of 8 in flight per GPU]"); `one_batch_in_flight` in the same JSON line is the strictly sequential
measurement (one batch of 8 owning the GPU: the number comparable with round 1 and with a latency reading of configs[1]),
`ms_per_step` is elapsed / steps (the throughput period, as the contract defines it) and `batch_latency_ms` says how long
one batch takes from its first kernel to its waveforms under each of the two arrangements.

Output: ONE JSON line on rank 0 with metric/value plus
  roofline     -- the dominant kernel (the implicit-GEMM engine): algorithmic FLOPs (2*M*N*K) of its launches / their
                  summed hipEvent durations, against the dense MFMA peak of the precision mode (157.3 TFLOP/s fp32,
                  2500 TFLOP/s bf16; bf16x3 spends 3 MFMAs per algorithmic multiply-add)
  cpu_baseline -- the CPU oracle (a port of the reference path) on this box's host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audiogpt_amd import config as C            # noqa: E402
from audiogpt_amd import weights as WT          # noqa: E402

# MI355X_MICROARCH.md: dense MFMA peaks.  The bf16x3 mode issues 3 bf16 MFMAs per algorithmic multiply-add
# (hi*hi + hi*lo + lo*hi), so its algorithmic ceiling is a third of the bf16 MFMA peak.
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}
MFMA_PER_FLOP = {"f32": 1, "bf16x3": 3, "bf16": 1}
CLIP_FRAMES = 624
LATENT = (4, 10, 78)
DDIM_STEPS = 100
CFG_SCALE = 1.5
PROMPTS_PER_GPU = 8


def synth_conditioning(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.layer_norm(torch.randn(n, 77, 1024, generator=g), (1024,))


def cpu_baseline(ddim_steps_sample=10):
    """Time the CPU oracle on this host: 1 latent with CFG, `ddim_steps_sample` of 100 DDIM steps (scaled; 10 = a whole
    BASELINE configs[0] job), plus one full VAE decode and one full HiFi-GAN pass.  Returns audio-seconds per
    second for one clip."""
    from oracle import ddim as O_ddim
    from oracle import unet as O_unet
    from oracle import vae as O_vae
    from oracle import vocoder as O_voc
    # a bounded thread count: on a many-core host torch's intra-op pool oversubscribes badly past ~32 threads
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    usd = WT.make_unet_state_dict(C.UNET_T2A, seed=0)
    vsd = WT.make_vae_state_dict(C.VAE_DDCONFIG, seed=1, with_encoder=False)
    gsd = O_voc.fold_weight_norm(WT.make_vocoder_state_dict(C.HIFIGAN_16K, seed=2))
    x = torch.from_numpy(np.random.RandomState(55).randn(1, *LATENT)).float()
    c, uc = synth_conditioning(1, 1234), synth_conditioning(1, 1235)
    ac = O_ddim.alphas_cumprod(1000, C.LDM_T2A["linear_start"], C.LDM_T2A["linear_end"])
    steps = O_ddim.ddim_timesteps(DDIM_STEPS)
    a, ap, sg, som = O_ddim.ddim_tables(ac, steps)
    with torch.no_grad():
        O_unet.unet_forward(usd, C.UNET_T2A, torch.cat([x, x]), torch.tensor([991, 991]), torch.cat([uc, c]))  # warm-up
        t0 = time.perf_counter()
        for i in range(ddim_steps_sample):
            idx = DDIM_STEPS - 1 - i
            ts = torch.full((2,), int(steps[idx]), dtype=torch.long)
            e_u, e_c = O_unet.unet_forward(usd, C.UNET_T2A, torch.cat([x, x]), ts, torch.cat([uc, c])).chunk(2)
            x, _ = O_ddim.ddim_step(x, e_u + CFG_SCALE * (e_c - e_u), a[idx], ap[idx], sg[idx], som[idx])
        t_unet = (time.perf_counter() - t0) / ddim_steps_sample
        t0 = time.perf_counter()
        mel = O_vae.decode_first_stage(vsd, C.VAE_DDCONFIG, x, 1.0)
        t_vae = time.perf_counter() - t0
        spec = torch.clamp((mel + 1.0) / 2.0, 0.0, 1.0)[:, 0]
        t0 = time.perf_counter()
        O_voc.hifigan_forward(gsd, C.HIFIGAN_16K, spec)
        t_voc = time.perf_counter() - t0
    clip_s = CLIP_FRAMES * 256 / 16000.0
    total = DDIM_STEPS * t_unet + t_vae + t_voc
    return dict(value=clip_s / total, unit="audio-seconds/sec", cores=cores, kind="port",
                sample="1 prompt: %d of %d CFG DDIM steps timed and scaled (%.2f s/step), + full VAE decode (%.2f s) "
                       "+ full HiFi-GAN 624 frames (%.2f s); torch %s fp32, %d threads"
                       % (ddim_steps_sample, DDIM_STEPS, t_unet, t_vae, t_voc, torch.__version__, cores),
                # the reference's own classes timed beside this port on the same 8 cores (the reference tree does not travel to the
                # GPU box): port time / reference time on the 10-step configs[0] job, two runs -- the port is bit-identical in
                # latent and mel and takes 0.75 - 0.94 x the reference's time, i.e. this baseline slightly flatters the CPU
                reference_time_ratio={"port_over_reference": [0.75, 0.942], "at_100_steps": [0.70, 0.973],
                                      "source": "profiles/r5_cpu_reference_vs_port.txt"},
                parts={"unet_cfg_step_s": t_unet, "vae_decode_s": t_vae, "hifigan_624_s": t_voc})


def cpu_baseline_mixed(ddim_steps_sample=2, S=DDIM_STEPS):
    """The CPU oracle on one clip of each tool of the mixed batch: `ddim_steps_sample` DDIM steps timed and scaled, the VAE
    and BigVGAN passes in full.  Returns audio-seconds per second for (one inpaint clip + one image-to-audio clip)."""
    from oracle import ddim as O_ddim
    from oracle import unet as O_unet
    from oracle import vae as O_vae
    from oracle import vocoder as O_voc
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    vsd = WT.make_vae_state_dict(C.VAE_DDCONFIG, seed=1, with_encoder=True)
    gsd = O_voc.fold_weight_norm(WT.make_vocoder_state_dict(C.BIGVGAN_16K, seed=3))
    g = torch.Generator().manual_seed(77)
    parts = {}
    with torch.no_grad():
        # inpaint: encode the masked mel, concat-conditioned DDIM without CFG, decode, BigVGAN over 848 frames
        usd = WT.make_unet_state_dict(C.UNET_INPAINT, seed=5)
        mel = torch.rand(1, 1, 80, 848, generator=g)
        t0 = time.perf_counter()
        mean, logvar = O_vae.encode_moments(vsd, C.VAE_DDCONFIG, mel * 2 - 1)
        parts["inpaint_encode"] = time.perf_counter() - t0
        x = torch.randn(1, 4, 10, 106, generator=g)
        cc = torch.cat((mean, torch.ones(1, 1, 10, 106)), dim=1)
        ts = torch.full((1,), 991, dtype=torch.long)
        O_unet.unet_forward(usd, C.UNET_INPAINT, torch.cat([x, cc], 1), ts, None)
        t0 = time.perf_counter()
        for _ in range(ddim_steps_sample):
            O_unet.unet_forward(usd, C.UNET_INPAINT, torch.cat([x, cc], 1), ts, None)
        parts["inpaint_unet_step"] = (time.perf_counter() - t0) / ddim_steps_sample
        t0 = time.perf_counter()
        m = O_vae.decode_first_stage(vsd, C.VAE_DDCONFIG, x, 1.0)
        parts["inpaint_decode"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        O_voc.bigvgan_forward(gsd, C.BIGVGAN_16K, torch.clamp((m + 1) / 2, 0, 1)[:, 0])
        parts["inpaint_bigvgan"] = time.perf_counter() - t0
        # image-to-audio: CFG 3 over a one-token context
        usd = WT.make_unet_state_dict(C.UNET_I2A, seed=4)
        x = torch.randn(1, *LATENT, generator=g)
        ctx2 = torch.randn(2, 1, 1024, generator=g)
        ts = torch.full((2,), 991, dtype=torch.long)
        O_unet.unet_forward(usd, C.UNET_I2A, torch.cat([x, x]), ts, ctx2)
        t0 = time.perf_counter()
        for _ in range(ddim_steps_sample):
            O_unet.unet_forward(usd, C.UNET_I2A, torch.cat([x, x]), ts, ctx2)
        parts["i2a_unet_step"] = (time.perf_counter() - t0) / ddim_steps_sample
        t0 = time.perf_counter()
        m = O_vae.decode_first_stage(vsd, C.VAE_DDCONFIG, x, 1.0)
        parts["i2a_decode"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        O_voc.bigvgan_forward(gsd, C.BIGVGAN_16K, torch.clamp((m + 1) / 2, 0, 1)[:, 0])
        parts["i2a_bigvgan"] = time.perf_counter() - t0
    total = S * (parts["inpaint_unet_step"] + parts["i2a_unet_step"]) + sum(v for k, v in parts.items() if "unet" not in k)
    audio = (848 + 624) * 256 / 16000.0
    return dict(value=audio / total, unit="audio-seconds/sec", cores=cores, kind="port",
                sample="1 inpaint clip + 1 image-to-audio clip: %d of %d DDIM steps of each UNet timed and scaled, VAE encode / decode "
                       "and BigVGAN in full (seconds: %s); torch %s fp32, %d threads"
                       % (ddim_steps_sample, S, ", ".join("%s %.2f" % kv for kv in parts.items()), torch.__version__, cores))


HIFIGAN64 = dict(B=64, T=1024, seed=7)


def hifigan64_mel(B=HIFIGAN64["B"], T=HIFIGAN64["T"], seed=HIFIGAN64["seed"]):
    """BASELINE.md section 2, config 3 (the same formula as tests/golden/make_golden.py hifigan_case)."""
    g = torch.Generator().manual_seed(seed)
    return torch.clamp(torch.randn(B, 80, T, generator=g) * 1.5 - 2.25, -6.0, 1.5)


class BoxSampler:
    """Best-effort record of what the GPU ran at during the timed region: a thread reads the amdgpu sysfs nodes of one card
    (current shader clock level of pp_dpm_sclk, socket power of its hwmon) twice a second.  Box-to-box spread of one binary is
    several percent and follows the clock a box sustains under this load; nothing here is required --
    every failure yields None."""

    def __init__(self, device=None, root="/sys/class/drm", period=0.5, pci_root="/sys/bus/pci/devices"):
        """device: the torch device the benchmark runs on.  Its PCI address (domain:bus:device.function of the HIP device, from
        torch.cuda.get_device_properties / hipDeviceGetPCIBusId) selects the sysfs node; only when that cannot be resolved does
        the sampler fall back to the first amdgpu card it finds, and says so in `matched_by`."""
        import glob
        import threading
        self.sclk, self.power, self.period = [], [], period
        self.other = {"mclk": [], "fclk": [], "socclk": []}      # memory / fabric / SoC clock levels, where the driver exposes them
        self._stop = threading.Event()
        self._thread = None
        self.card = None
        self.bdf = self.pci_bdf(device)
        self.matched_by = None
        if self.bdf:
            cand = os.path.join(pci_root, self.bdf)
            if os.path.exists(os.path.join(cand, "pp_dpm_sclk")):
                self.card, self.matched_by = cand, "pci_bus_id"
            else:       # the same device through its DRM node (containers that hide /sys/bus/pci)
                for c in sorted(glob.glob(os.path.join(root, "card[0-9]*"))):
                    try:
                        if os.path.basename(os.path.realpath(os.path.join(c, "device"))) == self.bdf and \
                                os.path.exists(os.path.join(c, "device", "pp_dpm_sclk")):
                            self.card, self.matched_by = os.path.join(c, "device"), "drm_node_of_pci_bus_id"
                            break
                    except OSError:
                        continue
        if self.card is None:
            for c in sorted(glob.glob(os.path.join(root, "card[0-9]*"))):
                if os.path.exists(os.path.join(c, "device", "pp_dpm_sclk")):
                    self.card, self.matched_by = os.path.join(c, "device"), "first_amdgpu_card (PCI address of the HIP device not resolved)"
                    break
        self._hw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*"))) if self.card else []

    @staticmethod
    def pci_bdf(device):
        """'dddd:bb:dd.f' of a torch CUDA(HIP) device, or None."""
        if device is None:
            return None
        try:
            pr = torch.cuda.get_device_properties(device)
            dom, bus, dv = (getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
            if bus is not None and dv is not None:
                return "%04x:%02x:%02x.0" % (int(dom or 0), int(bus), int(dv))
        except Exception:
            pass
        try:      # older torch: ask the HIP runtime
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
            if hip.hipDeviceGetPCIBusId(buf, 64, int(idx)) == 0:
                return buf.value.decode().lower()
        except Exception:
            pass
        return None

    @staticmethod
    def parse_sclk(text):
        """MHz of the level pp_dpm_sclk marks with '*' (None if there is none)."""
        import re
        for line in text.splitlines():
            if line.rstrip().endswith("*"):
                m = re.search(r"(\d+)\s*mhz", line.lower())
                if m:
                    return int(m.group(1))
        return None

    def sample(self):
        try:
            v = self.parse_sclk(open(os.path.join(self.card, "pp_dpm_sclk")).read())
            if v is not None:
                self.sclk.append(v)
        except Exception:
            pass
        for k, v in self.other.items():
            try:
                c = self.parse_sclk(open(os.path.join(self.card, "pp_dpm_" + k)).read())
                if c is not None:
                    v.append(c)
            except Exception:
                pass
        for h in self._hw:
            for f in ("power1_average", "power1_input"):
                try:
                    self.power.append(int(open(os.path.join(h, f)).read().strip()) / 1e6)     # microwatts
                    return
                except Exception:
                    continue

    def __enter__(self):
        import threading
        if self.card:
            def loop():
                while not self._stop.wait(self.period):
                    self.sample()
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        return False

    def summary(self):
        def med(v):
            return sorted(v)[len(v) // 2] if v else None
        return {"sclk_mhz_median": med(self.sclk), "sclk_mhz_min": min(self.sclk) if self.sclk else None,
                "mclk_mhz_median": med(self.other["mclk"]), "fclk_mhz_median": med(self.other["fclk"]),
                "socclk_mhz_median": med(self.other["socclk"]),
                "socket_power_w_median": med(self.power), "socket_power_w_max": max(self.power) if self.power else None,
                "samples": len(self.sclk), "pci_bus_id": self.bdf, "matched_by": self.matched_by,
                "source": "amdgpu sysfs (pp_dpm_sclk / mclk / fclk / socclk, hwmon power) of %s, sampled during the timed region" % self.card}


# What the calibration reads reach on the boxes that gave the fast-class numbers (profiles/README.md: 23.4 TB/s out of L2,
# 6.5 - 6.6 TB/s out of the Infinity Cache, 5.2 - 5.4 TB/s copy).  A box is put in the slow class when a read falls below 85 % of
# that: the kernels that lose on such boxes are the L2 -> LDS-bound ones, which neither the MFMA loop nor the
# copy loop tells apart.
BOX_CLASS_REF = {"l2_read_gbs": 23400.0, "infinity_cache_read_gbs": 6500.0, "copy_gbs": 5200.0, "mfma_bf16_tflops": 2250.0}


def box_class(calib, frac=0.85):
    """'fast' or 'slow(<which reads are low>)' from box.calib; None if the calibration did not run."""
    if not isinstance(calib, dict) or "error" in calib:
        return None
    low = [k for k, ref in BOX_CLASS_REF.items() if isinstance(calib.get(k), (int, float)) and calib[k] < frac * ref]
    return "fast" if not low else "slow(%s)" % ",".join(low)


def roofline_of(rows, precision):
    """Roofline object of the dominant implicit-GEMM kernel out of a maa_prof table (hipEvents on the library's stream)."""
    total_ms = sum(r["ms"] for r in rows.values())
    ig = {k: v for k, v in rows.items() if k.startswith("igemm")}
    dom = max(ig, key=lambda k: ig[k]["ms"])
    ig_ms = sum(v["ms"] for v in ig.values())
    ig_fl = sum(v["flops"] for v in ig.values())
    d = ig[dom]
    peak = PEAK_TFLOPS[precision]
    per = MFMA_PER_FLOP[precision] if "bf16" in dom else 1
    if "f32" in dom:
        peak = PEAK_TFLOPS["f32"]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    return {
        "bound": "mfma", "kernel": dom,
        "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
        "mfma_ops_per_algorithmic_flop": per, "frac_of_mfma_issue_peak": ach * per / peak,
        "traffic": None, "traffic_note": None,
        "launches": d["launches"], "avg_launch_us": 1e3 * d["ms"] / d["launches"],
        "flops_per_launch_avg": d["flops"] / d["launches"],
        "all_igemm": {"achieved": ig_fl / (ig_ms * 1e-3) / 1e12, "ms": ig_ms, "tflop": ig_fl / 1e12,
                      "share_of_kernel_time": ig_ms / total_ms},
        "kernel_time_ms": {k: round(v["ms"], 3) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])},
    }


def attach_traffic(roof, precision, section=None, units=None, table_path=None):
    """HBM-side bytes per launch (and MFMA-busy) of a roofline's dominant kernel from profiles/pmc_traffic.json -- measured by
    scripts/gpu_profile*.sh with rocprofv3 PMC passes on the GPU box right before the bench.  Accepted only if taken on THIS
    binary and launch mix: same sources (hash), same precision, and the same number of launches of that kernel per unit of
    work (`units` of this run: DDIM steps of the headline batch, generator passes / DDIM steps of a secondary workload)."""
    tpath = table_path or os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return
    from audiogpt_amd.build import _source_hash
    with open(tpath) as f:
        t = json.load(f)
    if section is not None:
        t = t.get("secondary", {}).get(section) or {}
    e = t.get("kernels", {}).get(roof["kernel"])
    mine = roof["launches"] / float(units)
    per = None if not e else e.get("launches_per_ddim_step", e.get("launches_per_unit"))
    if e and t.get("precision") == precision and t.get("source_hash") == _source_hash() and per and abs(per - mine) <= 0.03 * mine:
        roof["traffic"] = e["hbm_bytes_per_launch"]
        roof["traffic_source"] = "profiles/pmc_traffic.json"
        roof["traffic_note"] = ("NOT measured in this run: `traffic` / `mfma_busy` are attached from profiles/pmc_traffic.json (rocprofv3 PMC "
                                "passes of this same binary -- source hash, precision and launch mix matched -- taken on the builder's box). "
                                + t["note"])
        roof["mfma_busy"] = e.get("mfma_busy")      # SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles, same PMC call
    else:
        roof["traffic_note"] = "profiles/pmc_traffic.json does not match this binary / launch mix: not reported"


LINE_LIMIT = 6144          # the driver keeps the last 8.6 kB of stdout: the ONE JSON line must fit with room to spare
_ROOF_KEEP = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_mfma_issue_peak", "traffic", "mfma_busy",
              "traffic_source", "launches", "avg_launch_us")
_CPU_KEEP = ("value", "unit", "cores", "kind")
_PARITY_KEEP = ("mel_l1", "wav_rms", "gate", "meets_gate")


def _sig(v, n=5):
    """Floats to n significant digits (the detail file keeps full precision)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float("%.*g" % (n, v))


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if k in (d or {}) and d[k] is not None} if d else None


def slim_workload(r, top=False):
    """The part of one workload's record that goes into the stdout line: value / ms_per_step / dtype / config.workload and the
    three attachments (roofline, cpu_baseline, parity) cut down to their numbers.  Everything else -- per-kernel time tables,
    traffic notes, sample descriptions -- stays in the detail file."""
    if "error" in r:
        return {"error": r["error"][:160]}
    o = {k: _sig(r[k]) for k in ("value", "unit", "ms_per_step", "dtype", "steps") if k in r}
    if not top:
        o["workload"] = str((r.get("config") or {}).get("workload", ""))[:100]
    if r.get("one_batch_in_flight"):
        o["one_batch_in_flight"] = _pick(r["one_batch_in_flight"], ("value", "ms_per_step"))
    for k in ("T2A_txt2audio", "I2A_img2audio"):
        if k in r:
            o[k] = _pick(r[k], ("ms", "clip_seconds", "realtime_factor"))
    if r.get("roofline"):
        o["roofline"] = _pick(r["roofline"], _ROOF_KEEP)
        wp = r["roofline"].get("whole_pass")
        if wp:
            o["roofline"]["whole_pass_frac"] = _sig(wp["frac_of_mfma_peak"])
    if r.get("cpu_baseline"):
        o["cpu_baseline"] = _pick(r["cpu_baseline"], _CPU_KEEP)
    if r.get("parity") and _pick(r["parity"], _PARITY_KEEP):
        o["parity"] = _pick(r["parity"], _PARITY_KEEP)
    return o


def slim_line(result, detail_path=None, limit=LINE_LIMIT):
    """The ONE stdout JSON line (<= `limit` bytes) out of the full result record.  Top level: the driver's contract fields, the
    headline's roofline / cpu_baseline (with a one-line `sample`), `one_batch_in_flight` (BASELINE configs[1] literally: one
    batch of 8 owning the GPU), `one_batch_two_streams`, `box` with its calibration reads, and per secondary workload a
    slim_workload record.  `detail` names the file that holds the full record."""
    o = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in result}
    for k in ("value", "ms_per_step"):
        o[k] = _sig(o[k], 7)
    cfg = dict(result.get("config") or {})
    if "workload" in cfg:
        cfg["workload"] = cfg["workload"][:230]
    o["config"] = cfg
    for k in ("comm_ms_per_step", "batch_latency_ms"):
        if k in result:
            o[k] = {a: _sig(b) for a, b in result[k].items()}
    if result.get("roofline"):
        o["roofline"] = _pick(result["roofline"], _ROOF_KEEP)
        ai = result["roofline"].get("all_igemm")
        if ai:
            o["roofline"]["all_igemm_tflops"] = _sig(ai["achieved"])
            o["roofline"]["all_igemm_share"] = _sig(ai["share_of_kernel_time"])
        wp = result["roofline"].get("whole_pass")
        if wp:
            o["roofline"]["whole_pass_frac"] = _sig(wp["frac_of_mfma_peak"])
        kt = result["roofline"].get("kernel_time_ms") or {}
        tot = sum(kt.values()) or 1.0
        o["roofline"]["top_kernel_share"] = {k[:40]: _sig(v / tot, 3) for k, v in list(kt.items())[:5]}
    if result.get("cpu_baseline"):
        o["cpu_baseline"] = _pick(result["cpu_baseline"], _CPU_KEEP)
        o["cpu_baseline"]["sample"] = str(result["cpu_baseline"].get("sample", ""))[:160]
        if result["cpu_baseline"].get("reference_time_ratio"):
            o["cpu_baseline"]["reference_time_ratio"] = result["cpu_baseline"]["reference_time_ratio"]["port_over_reference"]
    if result.get("one_batch_in_flight"):
        o["one_batch_in_flight"] = _pick(result["one_batch_in_flight"], ("value", "ms_per_step", "steps", "cfg_lanes"))
    for k in ("one_batch_other_form", "one_batch_two_streams"):      # (the second: records of rounds 3 / 4)
        if result.get(k):
            o[k] = _pick(result[k], ("value", "ms_per_step", "cfg_lanes", "bit_identical", "bit_identical_to_one_stream"))
    if result.get("box"):
        b = result["box"]
        o["box"] = _pick(b, ("sclk_mhz_median", "mclk_mhz_median", "fclk_mhz_median", "socket_power_w_median", "pci_bus_id", "class"))
        if isinstance(b.get("calib"), dict):
            o["box"]["calib"] = {k: _sig(v) for k, v in b["calib"].items() if isinstance(v, (int, float))}
    for k in ("ranks_seen", "per_rank_value", "last_gather_shape"):
        if k in result:
            o[k] = result[k]
    if "secondary" in result:
        o["secondary"] = {k: slim_workload(v) for k, v in result["secondary"].items()}
    if detail_path:
        o["detail"] = detail_path
    line = json.dumps(o, separators=(",", ":"))
    if len(line) > limit:          # never exceed the capture: drop the optional parts, most verbose first
        for path in (("roofline", "top_kernel_share"), ("cpu_baseline", "sample"), ("config", "workload"), ("data",)):
            d = o
            for k in path[:-1]:
                d = d.get(k) or {}
            d.pop(path[-1], None)
            line = json.dumps(o, separators=(",", ":"))
            if len(line) <= limit:
                break
    assert len(line) <= limit, "bench line is %d bytes (> %d)" % (len(line), limit)
    return line


def emit(result, args):
    """Full record -> gpurun_out/bench_detail.json (and --json-out), slim line -> stdout."""
    detail_rel = os.path.join("gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, detail_rel), "w") as f:
            json.dump(result, f)
    except OSError:
        detail_rel = None
    if getattr(args, "json_out", None):
        with open(args.json_out, "w") as f:
            f.write(json.dumps(result))
    sys.stderr.write("[bench] full record (per-kernel tables, notes, samples): %s\n" % (detail_rel or "not written"))
    print(json.dumps(result) if getattr(args, "full_line", False) else slim_line(result, detail_rel), flush=True)


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


def _t2a_inputs(n, dev):
    x_T = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float().to(dev)
    return x_T, synth_conditioning(n, 1234).to(dev), synth_conditioning(1, 1235).to(dev).expand(n, -1, -1).contiguous()


def _timed(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k, out


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
    pipes[0].ctx.set_cfg_split(True)
    gen(pipes[0])                                  # (the step graph of the two-lane form)
    one, (wav, spec, z) = _timed(lambda: gen(pipes[0]), 2)
    pipes[0].ctx.set_cfg_split(inflight == 1)
    audio_s = pipes[0].audio_seconds(n, CLIP_FRAMES)
    res = {"metric": "generated audio-seconds/sec (10s clip, 100 DDIM steps) [%d independent batches of %d prompts in flight]" % (inflight, n),
           "value": audio_s / per_step, "unit": "audio-seconds/sec", "n_gpus": 1, "steps": steps, "warmup": 1,
           "ms_per_step": 1e3 * per_step, "higher_is_better": True, "dtype": precision,
           "data": "synthetic prompts (layer-normed N(0,1) [B,77,1024]); seeded random-init weights",
           "config": {"workload": label, "prompts_per_gpu": n, "ddim_steps": S, "batches_in_flight": inflight,
                      "audio_seconds_per_step": audio_s},
           "one_batch_in_flight": {"value": audio_s / one, "ms_per_step": 1e3 * one}}
    if roofline:
        pipes[0].ctx.prof_begin()
        pipes[0].generate(x_T, c, uc, CFG_SCALE, S, use_graph=False)
        res["roofline"] = roofline_of(pipes[0].ctx.prof_end(), precision)
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


class _StubPipe:
    """CPU stand-in for a MakeAnAudio replica (--stub-cpu: the N > 1 control flow of this file under gloo, tests/test_shard_gloo.py):
    a deterministic per-sample function of (x_T, c, uc) with the pipeline's output shapes in miniature."""
    stream = None
    ctx = None

    def generate_here(self, x_T, c, uc, scale, S, use_graph=True):
        feat = (c.mean(dim=(1, 2)) - uc.mean(dim=(1, 2)))[:, None] * scale + x_T.reshape(x_T.shape[0], -1).sum(dim=1, keepdim=True)
        wav = torch.sin(feat * 0.01 + torch.arange(64, dtype=torch.float32)[None, :] * 0.1)
        return wav, None, None

    generate = generate_here

    def audio_seconds(self, n, frames):
        return n * frames * 256 / 16000.0

    def close(self):
        pass


class _NullEvent:
    """torch.cuda.Event's surface on the CPU path."""

    def __init__(self, enable_timing=False):
        pass

    def record(self, stream=None):
        pass

    def elapsed_time(self, other):
        return 0.0


def self_launch(n, argv):
    """`python bench.py --gpus N ...` started without a launcher: re-run this file as N ranks under torch.distributed.run
    (--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1, a free port), pass the ranks' stdout / stderr through -- rank 0 prints
    the ONE JSON line -- and exit with the launcher's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ddim-steps", type=int, default=DDIM_STEPS)
    ap.add_argument("--prompts-per-gpu", type=int, default=PROMPTS_PER_GPU)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"],
                    help="contraction arithmetic: exact fp32 MFMA, bf16x3 split (default; meets the fp32 parity gates), bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table of one profiled batch to stderr")
    ap.add_argument("--workload", default="t2a", choices=["t2a", "hifigan64", "mixed"],
                    help="t2a: BASELINE configs[1] (the headline line, with the others under 'secondary'); hifigan64: configs[2] "
                         "alone; mixed: configs[4] on one GPU (inpaint + image-to-audio)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads of the default run")
    ap.add_argument("--secondary-only", default="", help="comma list: run only these secondary workloads (hifigan64, mixed, t2a_bf16, "
                                                         "t2a_bigvgan, tool_latency)")
    ap.add_argument("--legacy-streams", action="store_true",
                    help="replicas on library-created blocking streams (ordered against PyTorch's default stream) instead of one "
                         "private torch stream each")
    ap.add_argument("--stagger-ms", type=float, default=0.0,
                    help="replica k waits k x this long before its first batch of a run (A/B: do the replicas' DDIM steps overlap "
                         "better out of phase?)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="prompt batches in flight per GPU: consecutive steps (independent batches of 8 prompts) run on this many "
                         "pipeline replicas / HIP streams, as a serving loop would overlap requests; 1 = strictly one after another")
    ap.add_argument("--cfg-split", default="auto", choices=["auto", "0", "1"],
                    help="classifier-free guidance inside the sampler: 1 = the two halves of a step as two lanes (branches of the "
                         "captured step graph), 0 = one stream, auto = lanes only when ONE batch is in flight (--inflight 1): with "
                         "several replicas the chip is already full from outside and six concurrent lanes lose 24 %")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise the process group and issue C1 scatter / broadcast, C2 gather, the barriers and ranks_seen even "
                         "with ONE rank (RCCL exercised on the one GPU a test box has: tests/test_gpu_rccl.py)")
    ap.add_argument("--stub-cpu", action="store_true", help=argparse.SUPPRESS)      # tests: this file's control flow on CPU / gloo
    ap.add_argument("--json-out", default=None, help="also write the FULL record (what gpurun_out/bench_detail.json holds) to this file")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full record on stdout instead of the <= 6 kB line (per-kernel tables included: ~25 kB)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: become the launcher (one rank per GPU under torch.distributed.run on
        # 127.0.0.1, the command the contract names) and hand rank 0's line through; the ranks see WORLD_SIZE and take the path below
        return self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): use --nproc-per-node == --gpus (or run "
                         "`python bench.py --gpus N` without a launcher: it starts its own ranks)" % (args.gpus, world))
    stub = args.stub_cpu
    if stub:
        args.no_roofline = args.no_cpu_baseline = args.no_secondary = True
        args.workload = "t2a"
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU visible"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    Event = _NullEvent if stub else torch.cuda.Event
    if args.workload == "hifigan64":
        assert world == 1, "the vocoder-only workload is a single-GPU configuration"
        emit(run_hifigan64(dev, args.precision, args.steps, args.warmup, not args.no_cpu_baseline, not args.no_roofline), args)
        return
    if args.workload == "mixed":
        assert world == 1, "run one mixed batch per GPU (replicas) -- no collective in this workload"
        emit(run_mixed(dev, args.precision, args.steps, args.warmup, args.prompts_per_gpu, args.ddim_steps,
                       not args.no_roofline, not args.no_cpu_baseline), args)
        return
    dist = None
    force = args.force_collectives
    if world > 1 or force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from concurrent.futures import ThreadPoolExecutor

    from audiogpt_amd.pipeline import MakeAnAudio
    from audiogpt_amd.shard import broadcast_conditioning, gather_waveforms, ranks_seen, run_in_flight, start_codes
    # One batch of 8 prompts leaves much of the chip idle (its kernels are short and latency-bound: two independent
    # batches side by side finish in 1.57x the time of one, profiles/r2/r2_dual_stream_probe.txt), so consecutive steps of the
    # benchmark -- independent prompt batches, each sampled exactly as BASELINE configs[1] says -- are kept `inflight` at
    # a time on as many pipeline replicas (own HIP stream, workspace and weights), like a server overlapping requests.
    # Collectives stay on this thread, in step order.
    inflight = max(1, args.inflight)
    # each replica on its own (non-blocking) torch stream: with the library's default blocking streams every op on PyTorch's
    # legacy default stream -- the clamp between VAE and vocoder, the collectives' bookkeeping -- is a barrier across all
    # replicas (--legacy-streams keeps that arrangement for A/B runs)
    if stub:
        pipes = [_StubPipe() for _ in range(inflight)]
    else:
        pipes = [MakeAnAudio(dev, precision=args.precision, stream=None if args.legacy_streams else torch.cuda.Stream(dev))
                 for _ in range(inflight)]
    pipe = pipes[0]
    # CFG halves of a DDIM step as two lanes (the library's default; csrc/ddim.cpp): worth +4 % when ONE batch owns the GPU, but
    # with three batches in flight the chip is already full from outside and six concurrent lanes lose 24 % (profiles/
    # r5_call1_cfg_lanes_ab.txt) -- so the replicas of the overlapped arrangement run their steps on one stream each, as a
    # server that overlaps requests would configure them, and `one_batch_in_flight` runs with the lanes
    lanes = args.cfg_split == "1" or (args.cfg_split == "auto" and inflight == 1)
    lanes_one = args.cfg_split != "0"
    if not stub:
        for p_ in pipes:
            p_.ctx.set_cfg_split(lanes)
    # worker threads start with torch's thread-local device at 0: pin them to this rank's GPU (no stray context on GPU 0)
    pool = ThreadPoolExecutor(max_workers=inflight) if stub else \
        ThreadPoolExecutor(max_workers=inflight, initializer=torch.cuda.set_device, initargs=(dev,))
    n = args.prompts_per_gpu
    S = args.ddim_steps
    use_graph = not args.no_graph

    # synthetic prompt batch: rank 0 "runs the text encoder" for every rank's prompts
    if rank == 0:
        c_all = synth_conditioning(n * world, 1234).to(dev)
        uc_row = synth_conditioning(1, 1235).to(dev)
    else:
        c_all, uc_row = None, None

    # C0: every rank regenerates the start codes of the whole job and keeps its block (nothing is sent); done once,
    # outside the timed loop, like the batch geometry every rank knows up front (no host round trips per batch)
    x_T = start_codes(55, n * world, LATENT, world, rank).to(dev)
    cond_shape, counts = (n * world, 77, 1024), [n] * world

    def make_generator(p_, k_=0):
        first = [args.stagger_ms > 0 and k_ > 0]

        def generate(c_, uc_, ready):
            """One prompt batch on replica p_ (worker thread): everything on the replica's own stream, after the event the
            main thread recorded behind this batch's conditioning; returns the waveforms and the event that marks them done."""
            done = Event()
            if first[0]:
                first[0] = False
                time.sleep(k_ * args.stagger_ms * 1e-3)
            if p_.stream is None:
                wav = p_.generate_here(x_T, c_, uc_, CFG_SCALE, S, use_graph=use_graph)[0]
                done.record(None if stub else torch.cuda.current_stream(dev))
            else:
                with torch.cuda.stream(p_.stream):
                    p_.stream.wait_event(ready)
                    for t in (c_, uc_):          # allocated on the main thread's stream, read on this one: tell the allocator
                        t.record_stream(p_.stream)
                    wav = p_.generate_here(x_T, c_, uc_, CFG_SCALE, S, use_graph=use_graph)[0]
                    done.record(p_.stream)
            return wav, done
        return generate

    comm_events = {"C1_broadcast": [], "C2_gather": []}      # (start, end) event pairs around the two collectives, per step

    def conditioning():
        e0, e1 = Event(enable_timing=True), Event(enable_timing=True)
        e0.record()
        c, uc = broadcast_conditioning(c_all, uc_row, n, dev, dist, shape=cond_shape, force=force)   # C1: RCCL scatter + bcast (no-op at N = 1)
        e1.record()
        comm_events["C1_broadcast"].append((e0, e1))
        ready = Event()
        ready.record()
        return c, uc, ready

    def gather(res):
        wav, done = res
        if not stub:
            cur = torch.cuda.current_stream()
            cur.wait_event(done)
            wav.record_stream(cur)
        e0, e1 = Event(enable_timing=True), Event(enable_timing=True)
        e0.record()
        out = gather_waveforms(wav, dist, counts=counts, force=force)                          # C2: gather to rank 0
        e1.record()
        comm_events["C2_gather"].append((e0, e1))
        return out

    def run_steps(k):
        """k steps; step i runs on pipeline i % inflight while the previous inflight-1 steps are still sampling
        (audiogpt_amd.shard.run_in_flight: collectives on this thread, in step order)."""
        outs = run_in_flight(k, [make_generator(p_, k_) for k_, p_ in enumerate(pipes)], conditioning, gather, pool)
        return outs[-1] if outs else None

    def barrier():
        if dist is not None:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    run_steps(args.warmup * inflight)      # W untimed steps on every replica (each sizes its workspace, builds its graphs)
    barrier()
    for v in comm_events.values():
        v.clear()
    # box calibration right before the timed region (rank 0): what a fixed MFMA loop and a fixed copy reach on this box now
    calib = None
    if rank == 0:
        try:
            calib = pipe.ctx.calib() if not stub else {"stub": 1.0}
            calib["note"] = ("csrc/calib.hip on the first replica's stream: dense bf16 MFMA loop (peak 2500 TFLOP/s at 2.4 GHz), "
                             "a 256 MiB float4 copy (read + written bytes), every workgroup re-reading its own 64 KiB (L2), "
                             "128 MiB re-read by the whole grid (Infinity Cache)")
        except Exception as e:      # never lose the line to the calibration
            calib = {"error": str(e)[:200]}
    barrier()      # EVERY rank (a collective): rank 0's calibration loops must not run into the other ranks' timed region
    sampler = BoxSampler(dev) if (rank == 0 and not stub) else None
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if sampler is not None:
        sampler.__exit__()
    per_rank_elapsed = [elapsed]
    if dist is not None:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_elapsed = [float(e.item()) for e in every]
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    seen = ranks_seen(dev, dist, force=force) if (world > 1 or force) else None

    # device time of the two collectives of a step (events on the main thread's stream, this rank): their share of a step is
    # what the first multi-GPU run should look at before anything else
    comm_ms = {k: (sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1)) for k, v in comm_events.items()}
    audio_s = pipe.audio_seconds(n * world, CLIP_FRAMES) * args.steps
    method = "" if inflight == 1 else " [%d independent batches of %d prompts in flight per GPU]" % (inflight, n)
    result = {
        "metric": "generated audio-seconds/sec (10s clip, 100 DDIM steps)" + method,
        "value": audio_s / elapsed, "unit": "audio-seconds/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic prompts (layer-normed N(0,1) [B,77,1024]); seeded random-init weights",
        "config": {"workload": "Make-An-Audio T2A batch=%d/GPU, %d DDIM steps, CFG %.1f, UNet+VAE+HiFi-GAN(16k), "
                               "%s" % (n, S, CFG_SCALE, {"f32": "fp32 (exact-f32 MFMA)", "bf16x3": "fp32 storage, bf16x3-split MFMA (hi/lo, fp32 accumulate; meets the fp32 parity gates)", "bf16": "fp32 storage, bf16 MFMA operands"}[args.precision]),
                   "prompts_per_gpu": n, "ddim_steps": S, "latent": list(LATENT), "mel_frames": CLIP_FRAMES,
                   "audio_seconds_per_step": pipe.audio_seconds(n * world, CLIP_FRAMES), "hipgraph": use_graph,
                   "batches_in_flight": inflight, "cfg_lanes": 2 if lanes else 1,
                   "parallelism": "prompt-sharded x%d (RCCL bcast cond / gather wav)" % world},
        "comm_ms_per_step": {k: round(v, 4) for k, v in comm_ms.items()},
        # Little's law for the overlapped arrangement: `inflight` batches are resident for `inflight` throughput periods
        "batch_latency_ms": {"in_flight": 1000.0 * elapsed / args.steps * inflight},
    }
    if sampler is not None:
        result["box"] = sampler.summary()
        if calib is not None:
            result["box"]["calib"] = calib
            result["box"]["class"] = box_class(calib)
    if world > 1 or force:
        # proof of what an N > 1 line ran on: the device identity of every rank (PCI address; must be N distinct ones) and each
        # rank's own rate over its own clock (the line's value uses the slowest rank's time)
        result["ranks_seen"] = seen
        result["per_rank_value"] = [pipe.audio_seconds(n, CLIP_FRAMES) * args.steps / e for e in per_rank_elapsed]

    if rank == 0 and not args.no_roofline:
        # one more batch, eager (graph launches cannot be event-timed), every kernel bracketed by hipEvents on the
        # library's stream; the dominant kernel family is the implicit-GEMM engine of the precision mode
        c = c_all[:n]
        uc = uc_row.expand(n, -1, -1).contiguous()
        pipe.ctx.prof_begin()
        pipe.generate(x_T, c, uc, CFG_SCALE, S, use_graph=False)
        rows = pipe.ctx.prof_end()
        result["roofline"] = roofline_of(rows, args.precision)
        # HBM-side traffic per launch: measured by scripts/gpu_profile.sh on the GPU box right before this run (separate
        # rocprofv3 PMC passes); accepted only for this binary and launch mix (attach_traffic)
        attach_traffic(result["roofline"], args.precision, None, S)
        if args.breakdown:
            for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
                sys.stderr.write("%-28s launches %6d  ms %10.3f  TFLOP/s %8.2f  GB/s %9.1f\n" % (
                    k, v["launches"], v["ms"], v["flops"] / max(v["ms"], 1e-9) / 1e9, v["bytes"] / max(v["ms"], 1e-9) / 1e6))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only (bounded sample, ~25 s of host time)
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0 and world == 1 and inflight > 1 and args.steps >= 2:
        # the same K steps strictly one batch after another on one stream (the latency-oriented number)
        k1 = min(args.steps, 3)
        if not stub:
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
    if rank == 0 and world == 1 and not args.no_secondary:
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
    if rank == 0:
        if out is not None:      # identity of the last step's gathered waveforms (bit-identity checks between arrangements)
            import hashlib
            result["wav_sha16"] = hashlib.sha256(out.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
        if stub:
            result["data"] = "stub pipeline on CPU (control-flow test): not a measurement"
            result["last_gather_shape"] = list(out.shape) if out is not None else None
        emit(result, args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
