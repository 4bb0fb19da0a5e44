"""Multi-process prompt sharding (C0/C1/C2 of audiogpt_amd/shard.py) on CPU with the gloo backend, world_size 2.
The per-rank "generation" is a deterministic stand-in (the HIP path needs a GPU); what is tested is that the
sharded job reproduces the single-process result bit for bit, including ragged prompt counts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audiogpt_amd import shard  # noqa: E402


def _fake_generate(x_T, c, uc):
    """Per-sample, batch-independent stand-in for DDIM+VAE+vocoder: wav[i] depends only on sample i."""
    feat = (c.mean(dim=(1, 2)) - uc.mean(dim=(1, 2)))[:, None] + x_T.flatten(1).sum(dim=1, keepdim=True)      # (rows may be 0: a rank without prompts)
    t = torch.arange(64, dtype=torch.float32)[None, :]
    return torch.sin(feat * 0.01 + t * 0.1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_path, known=False, mode="scatter"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    lo, hi = shard.shard_range(n_total, world, rank)
    if rank == 0:
        g = torch.Generator().manual_seed(1234)
        c_all = torch.randn(n_total, 7, 16, generator=g)
        uc_row = torch.randn(1, 7, 16, generator=g)
    else:
        c_all, uc_row = None, None
    # known = the batch geometry is agreed up front (bench.py's timed loop): no metadata / count exchange
    shape = (n_total, 7, 16) if known else None
    counts = [b - a for a, b in (shard.shard_range(n_total, world, r) for r in range(world))] if known else None
    c, uc = shard.broadcast_conditioning(c_all, uc_row, hi - lo, dev, dist, shape=shape, mode=mode)
    assert c.shape[0] == hi - lo
    x_T = shard.start_codes(55, n_total, (4, 2, 3), world, rank)
    wav = _fake_generate(x_T, c, uc)
    full = shard.gather_waveforms(wav, dist, counts=counts)
    seen = shard.ranks_seen(dev, dist)
    assert len(seen["ids"]) == world and seen["n_distinct"] == world, seen      # one process (here: pid) per rank
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,known,mode", [(2, 8, False, "scatter"), (2, 5, False, "scatter"), (2, 8, True, "scatter"),
                                                      (2, 5, True, "broadcast"), (3, 8, True, "scatter"), (3, 7, False, "broadcast"),
                                                      (8, 64, True, "scatter"), (8, 13, False, "scatter"), (8, 5, True, "scatter")])
def test_sharded_equals_single_process(tmp_path, world, n_total, known, mode):
    """C0 / C1 / C2 with 2, 3 and 8 ranks (BASELINE configs[3] is 8-way), even, ragged and fewer-prompts-than-ranks jobs, C1 as
    the per-peer scatter and as the round-1 broadcast."""
    out = str(tmp_path / "wav.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_total, out, known, mode), nprocs=world, join=True)
    got = np.load(out)
    g = torch.Generator().manual_seed(1234)
    c_all = torch.randn(n_total, 7, 16, generator=g)
    uc_row = torch.randn(1, 7, 16, generator=g)
    c, uc = shard.broadcast_conditioning(c_all, uc_row, n_total, torch.device("cpu"), None)
    ref = _fake_generate(shard.start_codes(55, n_total, (4, 2, 3)), c, uc).numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_start_codes_match_reference_recipe():
    """audio-chatgpt.py:160-162: RandomState(seed).randn(n, 4, H//8, W//8) -> float32."""
    x = shard.start_codes(55, 3, (4, 10, 78))
    ref = torch.from_numpy(np.random.RandomState(55).randn(3, 4, 10, 78)).to(torch.float32)
    assert torch.equal(x, ref)
    parts = [shard.start_codes(55, 3, (4, 10, 78), 2, r) for r in range(2)]
    assert torch.equal(torch.cat(parts), ref)


def _inflight_worker(rank, world, port, n_total, steps, inflight, out_path):
    """bench.py's timed loop on CPU: `steps` prompt batches, `inflight` of them generating at once on worker threads
    (with rank- and step-dependent delays so the threads finish out of order), collectives on the main thread."""
    import time
    from concurrent.futures import ThreadPoolExecutor

    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    lo, hi = shard.shard_range(n_total, world, rank)
    counts = [b - a for a, b in (shard.shard_range(n_total, world, r) for r in range(world))]
    x_T = shard.start_codes(55, n_total, (4, 2, 3), world, rank)
    state = {"step": 0}

    def conditioning():          # a different prompt batch per step; rank 0 "ran the text encoder"
        i = state["step"]
        state["step"] += 1
        if rank == 0:
            g = torch.Generator().manual_seed(1000 + i)
            c_all, uc_row = torch.randn(n_total, 7, 16, generator=g), torch.randn(1, 7, 16, generator=g)
        else:
            c_all, uc_row = None, None
        return shard.broadcast_conditioning(c_all, uc_row, hi - lo, dev, dist, shape=(n_total, 7, 16))

    def make_generator(slot):
        calls = {"n": 0}

        def generate(c, uc):
            calls["n"] += 1
            time.sleep(0.02 * ((slot * 3 + rank * 5 + calls["n"]) % 4))        # out-of-order completion
            return _fake_generate(x_T, c, uc)
        return generate
    pool = ThreadPoolExecutor(max_workers=inflight)
    outs = shard.run_in_flight(steps, [make_generator(s) for s in range(inflight)], conditioning,
                               lambda wav: shard.gather_waveforms(wav, dist, counts=counts), pool)
    assert len(outs) == steps
    if rank == 0:
        np.save(out_path, torch.stack(outs).numpy())
    else:
        assert all(o is None for o in outs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("steps,inflight", [(6, 3), (4, 3), (2, 3), (5, 1)])
def test_batches_in_flight_keep_the_collective_order(tmp_path, steps, inflight):
    """Every step's gathered waveforms equal the single-process result of THAT step's prompts (no cross-step mix-up, no
    deadlock), whatever order the generating threads finish in; ragged shards (7 prompts over 2 ranks)."""
    n_total = 7
    out = str(tmp_path / "wav.npy")
    mp.spawn(_inflight_worker, args=(2, _free_port(), n_total, steps, inflight, out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape[0] == steps
    x_T = shard.start_codes(55, n_total, (4, 2, 3))
    for i in range(steps):
        g = torch.Generator().manual_seed(1000 + i)
        c_all, uc_row = torch.randn(n_total, 7, 16, generator=g), torch.randn(1, 7, 16, generator=g)
        c, uc = shard.broadcast_conditioning(c_all, uc_row, n_total, torch.device("cpu"), None)
        assert np.array_equal(got[i], _fake_generate(x_T, c, uc).numpy()), i


def _bench_worker(rank, world, port, out_path, inflight):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    import bench
    bench.main(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--inflight", str(inflight), "--stub-cpu",
                "--json-out", out_path])


@pytest.mark.parametrize("inflight", [1, 3])
def test_bench_main_multi_rank_control_flow(tmp_path, inflight):
    """bench.py's own N > 1 path end to end under gloo with a stub pipeline (two ranks): process-group set-up, C0 start codes,
    C1 broadcast and C2 gather issued in step order around worker threads that keep `inflight` batches going, the barrier /
    max-over-ranks timing, and ONE JSON line on rank 0 that follows the contract (whole-job value over both ranks)."""
    import json
    world, out = 2, str(tmp_path / "line.json")
    mp.spawn(_bench_worker, args=(world, _free_port(), out, inflight), nprocs=world, join=True)
    d = json.load(open(out))
    # the N > 1 line proves what it ran on: one identity per rank, all distinct, and every rank's own rate
    assert len(d["ranks_seen"]["ids"]) == 2 and d["ranks_seen"]["n_distinct"] == 2
    assert len(d["per_rank_value"]) == 2 and all(v > 0 for v in d["per_rank_value"])
    assert d["value"] <= sum(d["per_rank_value"]) * (1 + 1e-9)      # max-over-ranks time: the whole job is no faster than its parts
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"]
    assert d["config"]["batches_in_flight"] == inflight and d["config"]["prompts_per_gpu"] == 8
    assert ("in flight" in d["metric"]) == (inflight > 1)
    # value = audio-seconds of ALL ranks' prompts per step / seconds per step
    assert abs(d["value"] - d["config"]["audio_seconds_per_step"] / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    assert abs(d["config"]["audio_seconds_per_step"] - 2 * 8 * 624 * 256 / 16000.0) < 1e-9
    assert d["last_gather_shape"] == [16, 64]                     # rank 0 ends up with both ranks' waveforms
    assert set(d["comm_ms_per_step"]) == {"C1_broadcast", "C2_gather"}
    assert "roofline" not in d and "cpu_baseline" not in d         # N > 1 lines carry neither (and the stub measures nothing)


def test_bench_py_starts_its_own_ranks_when_run_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2 ...` typed as is (no torchrun around it, WORLD_SIZE unset): bench.py re-runs itself as two ranks
    under torch.distributed.run on 127.0.0.1 and rank 0 prints the ONE JSON line of the whole job (the driver's
    N = 8 run must not die on a launcher assumption)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-cpu"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["ranks_seen"]["n_distinct"] == 2 and len(d["per_rank_value"]) == 2
    # a launcher that started the wrong number of ranks is an error message, not an assert
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-cpu"], capture_output=True, text=True,
                       env=env2, cwd=str(tmp_path), timeout=120)
    assert r.returncode != 0 and "--nproc-per-node == --gpus" in r.stderr
