"""Prompt sharding over the GPUs of one node (one process per GPU, torch.distributed; "nccl" is RCCL on ROCm).

The reference has no inference-time multi-GPU path (audio-chatgpt.py:1055-1072 pins a device string per tool);
this layer is new.  Every latent is an independent DDIM trajectory (GroupNorm / LayerNorm / attention are
per-sample), so prompts shard as contiguous blocks with full weight replicas and NO collective inside the DDIM
loop:
  C0  x_T: every rank regenerates RandomState(seed).randn(n_total, ...) and slices its block -- bit-identical
      to the single-GPU run, nothing is sent
  C1  conditioning: the rank that ran the text/image encoder SCATTERS c [n_total, L, 1024] -- every peer receives only its
      own block (2.5 MB of a 64-prompt job's 20 MB) -- and broadcasts the single unconditional row over xGMI
      (`mode="broadcast"` keeps the round-1 form: everything to everyone, each rank slices)
  C2  waveforms: gathered to rank 0
Works with any backend ("gloo" on CPU in the tests).
"""
import numpy as np
import torch


def shard_range(n_total, world, rank):
    """Contiguous block of prompt indices owned by `rank`; sizes differ by at most one (ragged batches)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def start_codes(seed, n_total, shape, world=1, rank=0):
    """C0: the tools' start_code (audio-chatgpt.py:160-162), sliced to this rank's prompts."""
    x = np.random.RandomState(seed).randn(n_total, *shape)
    lo, hi = shard_range(n_total, world, rank)
    return torch.from_numpy(x[lo:hi]).to(torch.float32)


def _single(dist, force):
    """True when the collective can be skipped: no process group, or a world of one (unless `force`: the world-size-1 RCCL
    test drives scatter / broadcast / gather / all_gather_object down their collective branches on the one GPU it has)."""
    if dist is None or not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and not force


def broadcast_conditioning(c_all, uc_row, n_local, device, dist=None, src=0, shape=None, mode="scatter", force=False):
    """C1.  c_all [n_total, L, D] and uc_row [1, L, D] exist on `src` (None elsewhere).
    Returns (c_local [n_local, L, D], uc_local [n_local, L, D]) on every rank.
    `shape` = (n_total, L, D) when every rank already knows it (a serving loop with fixed batch geometry): skips the
    metadata broadcast and its device -> host synchronisation.
    mode "scatter" (default): `src` sends rank r its own block only (one grouped send/recv on RCCL; blocks of ragged jobs
    are padded to the largest); "broadcast": the whole tensor to every rank, sliced locally."""
    if _single(dist, force):
        return c_all[:n_local].contiguous(), uc_row.expand(n_local, -1, -1).contiguous()
    world, rank = dist.get_world_size(), dist.get_rank()
    if shape is not None:
        n_total, L, D = (int(v) for v in shape)
    else:
        meta = torch.zeros(3, dtype=torch.int64, device=device)
        if rank == src:
            meta = torch.tensor([c_all.shape[0], c_all.shape[1], c_all.shape[2]], dtype=torch.int64, device=device)
        dist.broadcast(meta, src)
        n_total, L, D = (int(v) for v in meta.tolist())
    lo, hi = shard_range(n_total, world, rank)
    assert hi - lo == n_local, (lo, hi, n_local)
    if rank != src:
        uc_row = torch.empty(1, L, D, dtype=torch.float32, device=device)
    if mode == "broadcast":
        if rank != src:
            c_all = torch.empty(n_total, L, D, dtype=torch.float32, device=device)
        dist.broadcast(c_all, src)
        c_local = c_all[lo:hi].contiguous()
    else:
        spans = [shard_range(n_total, world, r) for r in range(world)]
        nmax = max(b - a for a, b in spans)
        recv = torch.empty(nmax, L, D, dtype=torch.float32, device=device)
        blocks = None
        if rank == src:
            blocks = []
            for a, b in spans:
                if b - a == nmax:
                    blocks.append(c_all[a:b].contiguous())
                else:       # ragged job: pad the short blocks (scatter moves equal-sized pieces)
                    pad = torch.zeros(nmax, L, D, dtype=torch.float32, device=device)
                    pad[: b - a] = c_all[a:b]
                    blocks.append(pad)
        dist.scatter(recv, blocks, src=src)
        c_local = recv[:n_local].contiguous()
    dist.broadcast(uc_row, src)
    return c_local, uc_row.expand(n_local, -1, -1).contiguous()


def ranks_seen(device, dist=None, force=False):
    """Identity of the device every rank runs on, gathered to all ranks: the first N > 1 run can check that RCCL really had N
    distinct GPUs (a mis-set HIP_VISIBLE_DEVICES puts every rank on one).  Returns {"ids": [...], "n_distinct": k}."""
    mine = device_identity(device)
    if _single(dist, force):
        return {"ids": [mine], "n_distinct": 1}
    ids = [None] * dist.get_world_size()
    dist.all_gather_object(ids, mine)
    return {"ids": ids, "n_distinct": len(set(ids))}


def device_identity(device):
    """'pci:dddd:bb:dd.f' (or the device UUID) of a CUDA/HIP device; 'cpu:<pid>' for the CPU stand-in of the tests."""
    import os
    device = torch.device(device)
    if device.type != "cuda":
        return "cpu:%d" % os.getpid()
    try:
        pr = torch.cuda.get_device_properties(device)
        bus, dv = getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)
        if bus is not None and dv is not None:
            return "pci:%04x:%02x:%02x.0" % (int(getattr(pr, "pci_domain_id", 0) or 0), int(bus), int(dv))
        uuid = getattr(pr, "uuid", None)
        if uuid is not None:
            return "uuid:%s" % uuid
    except Exception:
        pass
    return "cuda-index:%d" % (device.index if device.index is not None else torch.cuda.current_device())


def gather_waveforms(wav_local, dist=None, dst=0, counts=None, force=False):
    """C2.  wav_local [n_local, T] -> [n_total, T] on `dst` (None elsewhere); ragged shards allowed.
    `counts` = per-rank row counts when known up front (skips the count exchange and its host synchronisation)."""
    if _single(dist, force):
        return wav_local
    world, rank = dist.get_world_size(), dist.get_rank()
    if counts is None:
        cts = [torch.zeros(1, dtype=torch.int64, device=wav_local.device) for _ in range(world)]
        dist.all_gather(cts, torch.tensor([wav_local.shape[0]], dtype=torch.int64, device=wav_local.device))
        counts = [int(c.item()) for c in cts]
    assert len(counts) == world and counts[rank] == wav_local.shape[0], (counts, rank, wav_local.shape)
    nmax = max(counts)
    pad = torch.zeros(nmax, wav_local.shape[1], dtype=wav_local.dtype, device=wav_local.device)
    pad[: wav_local.shape[0]] = wav_local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def run_in_flight(steps, generators, conditioning, gather, pool):
    """`steps` independent prompt batches with up to len(generators) of them sampling at once (bench.py's timed loop; a
    server overlapping requests).  Batch i is generated by `generators[i % len(generators)](*conditioning())` on a worker thread of
    `pool`; the collectives either side of it -- `conditioning()` -> the generator's arguments (C1) and `gather(result)` (C2) -- are issued by
    the CALLING thread only, in step order, so every rank issues the same collective sequence whatever the threads do:
        C1(0) .. C1(m-1), then for i >= m-1:  [C1(i+1)], C2(i-m+1) ...   with m = len(generators).
    Returns the list of gather results (one per step, in step order)."""
    m = len(generators)
    futs, outs = [], []
    for i in range(steps):
        futs.append(pool.submit(generators[i % m], *conditioning()))
        if len(futs) >= m:
            outs.append(gather(futs.pop(0).result()))
    for f in futs:
        outs.append(gather(f.result()))
    return outs
