"""Throughput benchmark of the Make-An-Audio hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
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


def cpu_baseline(ddim_steps_sample=2):
    """Time the CPU oracle on this host: 1 latent with CFG, `ddim_steps_sample` of 100 DDIM steps (scaled),
    plus one full VAE decode and one full HiFi-GAN pass.  Returns audio-seconds per second for one clip."""
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
                       % (ddim_steps_sample, DDIM_STEPS, t_unet, t_vae, t_voc, torch.__version__, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ddim-steps", type=int, default=DDIM_STEPS)
    ap.add_argument("--prompts-per-gpu", type=int, default=PROMPTS_PER_GPU)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"],
                    help="contraction arithmetic: exact fp32 MFMA, bf16x3 split (default; meets the fp32 parity gates), bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table of one profiled batch to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py measures the HIP path; no GPU visible"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from audiogpt_amd.pipeline import MakeAnAudio
    from audiogpt_amd.shard import broadcast_conditioning, gather_waveforms
    pipe = MakeAnAudio(dev, precision=args.precision)
    n = args.prompts_per_gpu
    S = args.ddim_steps
    use_graph = not args.no_graph

    # synthetic prompt batch: rank 0 "runs the text encoder" for every rank's prompts
    if rank == 0:
        c_all = synth_conditioning(n * world, 1234).to(dev)
        uc_row = synth_conditioning(1, 1235).to(dev)
    else:
        c_all, uc_row = None, None

    def one_batch():
        c, uc = broadcast_conditioning(c_all, uc_row, n, dev, dist)          # C1: RCCL broadcast (no-op at N = 1)
        x_T = torch.from_numpy(np.random.RandomState(55).randn(n * world, *LATENT)[rank * n:(rank + 1) * n]).float().to(dev)
        wav, spec, z = pipe.generate(x_T, c, uc, CFG_SCALE, S, use_graph=use_graph)
        return gather_waveforms(wav, dist)                                   # C2: gather to rank 0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_batch()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_batch()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    audio_s = pipe.audio_seconds(n * world, CLIP_FRAMES) * args.steps
    result = {
        "metric": "generated audio-seconds/sec (10s clip, 100 DDIM steps)",
        "value": audio_s / elapsed, "unit": "audio-seconds/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic prompts (layer-normed N(0,1) [B,77,1024]); seeded random-init weights",
        "config": {"workload": "Make-An-Audio T2A batch=%d/GPU, %d DDIM steps, CFG %.1f, UNet+VAE+HiFi-GAN(16k), "
                               "%s" % (n, S, CFG_SCALE, {"f32": "fp32 (exact-f32 MFMA)", "bf16x3": "fp32 storage, bf16x3-split MFMA (hi/lo, fp32 accumulate; meets the fp32 parity gates)", "bf16": "fp32 storage, bf16 MFMA operands"}[args.precision]),
                   "prompts_per_gpu": n, "ddim_steps": S, "latent": list(LATENT), "mel_frames": CLIP_FRAMES,
                   "audio_seconds_per_step": pipe.audio_seconds(n * world, CLIP_FRAMES), "hipgraph": use_graph,
                   "parallelism": "prompt-sharded x%d (RCCL bcast cond / gather wav)" % world},
    }

    if rank == 0 and not args.no_roofline:
        # one more batch, eager (graph launches cannot be event-timed), every kernel bracketed by hipEvents on the
        # library's stream; the dominant kernel family is the implicit-GEMM engine of the precision mode
        x_T = torch.from_numpy(np.random.RandomState(55).randn(n, *LATENT)).float().to(dev)
        c = c_all[:n]
        uc = uc_row.expand(n, -1, -1).contiguous()
        pipe.ctx.prof_begin()
        pipe.generate(x_T, c, uc, CFG_SCALE, S, use_graph=False)
        rows = pipe.ctx.prof_end()
        total_ms = sum(r["ms"] for r in rows.values())
        ig = {k: v for k, v in rows.items() if k.startswith("igemm")}
        dom = max(ig, key=lambda k: ig[k]["ms"])
        ig_ms = sum(v["ms"] for v in ig.values())
        ig_fl = sum(v["flops"] for v in ig.values())
        d = ig[dom]
        peak = PEAK_TFLOPS[args.precision]
        per = MFMA_PER_FLOP[args.precision] if "bf16" in dom else 1
        if "f32" in dom:
            peak = PEAK_TFLOPS["f32"]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                t = json.load(f)
            if t.get("precision") == args.precision and t.get("kernel_family") == dom.split("<")[0]:
                traffic, traffic_note = t["hbm_bytes_per_launch"], t["note"]
        result["roofline"] = {
            "bound": "mfma", "kernel": dom,
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "mfma_ops_per_algorithmic_flop": per, "frac_of_mfma_issue_peak": ach * per / peak,
            "traffic": traffic, "traffic_note": traffic_note,
            "launches": d["launches"], "avg_launch_us": 1e3 * d["ms"] / d["launches"],
            "flops_per_launch_avg": d["flops"] / d["launches"],
            "all_igemm": {"achieved": ig_fl / (ig_ms * 1e-3) / 1e12, "ms": ig_ms, "tflop": ig_fl / 1e12,
                          "share_of_kernel_time": ig_ms / total_ms},
            "kernel_time_ms": {k: round(v["ms"], 3) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if args.breakdown:
            for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
                sys.stderr.write("%-28s launches %6d  ms %10.3f  TFLOP/s %8.2f  GB/s %9.1f\n" % (
                    k, v["launches"], v["ms"], v["flops"] / max(v["ms"], 1e-9) / 1e9, v["bytes"] / max(v["ms"], 1e-9) / 1e6))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N = 1 only (bounded sample, ~25 s of host time)
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
