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
from benchlib.box import BoxSampler, box_class      # noqa: E402
from benchlib.common import (CFG_SCALE, CLIP_FRAMES, DDIM_STEPS, HIFIGAN64, LATENT, MFMA_PER_FLOP, PEAK_TFLOPS,      # noqa: E402,F401
                             PROMPTS_PER_GPU, hifigan64_mel, synth_conditioning)
from benchlib.cpu import cpu_baseline, cpu_baseline_mixed      # noqa: E402,F401
from benchlib.launch import _NullEvent, _StubPipe, self_launch      # noqa: E402,F401
from benchlib.line import LINE_LIMIT, emit, slim_line, slim_workload      # noqa: E402,F401
from benchlib.roofline import attach_traffic, roofline_of      # noqa: E402,F401
from benchlib.secondary import (mixed_inpaint, mixed_inputs, one_batch_records, run_hifigan64, run_mixed,      # noqa: E402,F401
                                run_secondaries, run_t2a_variant, run_tool_latency)


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
                         "several replicas the chip is already full from outside and six concurrent lanes lose 24 %%")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise the process group and issue C1 scatter / broadcast, C2 gather, the barriers and ranks_seen even "
                         "with ONE rank (RCCL exercised on the one GPU a test box has: tests/test_gpu_rccl.py)")
    ap.add_argument("--concurrency", type=int, default=0,
                    help="the serving-arrangement hint handed to the library (maa_ctx_set_concurrency); 0 = --inflight.  Profile passes "
                         "run ONE stream with the headline's launches: --inflight 1 --cfg-split 0 --concurrency 3")
    ap.add_argument("--no-one-batch", action="store_true", help="skip the `one_batch_in_flight` measurement that follows an overlapped run")
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
            p_.ctx.set_concurrency(args.concurrency or inflight)      # the serving arrangement as a hint (tiles by total workgroup time when >= 3)
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
                   "concurrency_hint": args.concurrency or inflight,      # maa_ctx_set_concurrency: >= 3 -> short-K tiles by total workgroup time
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
        # (ONE stream: a launch's duration only means something with the tiles chosen for a launch alone, so the context is told it
        # owns the GPU for this pass -- the timed region's replicas take the short-K contractions' tiles by total workgroup time,
        # maa_ctx_set_concurrency; the convolution engine, the dominant kernel, is the same code in both arrangements)
        c = c_all[:n]
        uc = uc_row.expand(n, -1, -1).contiguous()
        pipe.ctx.set_concurrency(1)
        pipe.ctx.set_cfg_split(False)
        pipe.ctx.prof_begin()
        pipe.generate(x_T, c, uc, CFG_SCALE, S, use_graph=False)
        rows = pipe.ctx.prof_end()
        pipe.ctx.set_concurrency(args.concurrency or inflight)
        pipe.ctx.set_cfg_split(lanes)
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
    if rank == 0 and world == 1 and inflight > 1 and args.steps >= 2 and not args.no_one_batch:
        one_batch_records(result, pipe, x_T, c_all, uc_row, n, S, use_graph, lanes, lanes_one, stub, args.steps, barrier)
    if rank == 0 and world == 1 and not args.no_secondary:
        run_secondaries(result, pipes, dev, args)
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
