"""Prompt sharding over the GPUs of one node (one process per GPU, torch.distributed; "nccl" is RCCL on ROCm).

The reference has no inference-time multi-GPU path (audio-chatgpt.py:1055-1072 pins a device string per tool);
this layer is new.  Every latent is an independent DDIM trajectory (GroupNorm / LayerNorm / attention are
per-sample), so prompts shard as contiguous blocks with full weight replicas and NO collective inside the DDIM
loop:
  C0  x_T: every rank regenerates RandomState(seed).randn(n_total, ...) and slices its block -- bit-identical
      to the single-GPU run, nothing is sent
  C1  conditioning: the rank that ran the text/image encoder broadcasts c [n_total, L, 1024] (and the single
      unconditional row) over xGMI; each rank keeps its slice
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


def broadcast_conditioning(c_all, uc_row, n_local, device, dist=None, src=0, shape=None):
    """C1.  c_all [n_total, L, D] and uc_row [1, L, D] exist on `src` (None elsewhere).
    Returns (c_local [n_local, L, D], uc_local [n_local, L, D]) on every rank.
    `shape` = (n_total, L, D) when every rank already knows it (a serving loop with fixed batch geometry): skips the
    metadata broadcast and its device -> host synchronisation."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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
    if rank != src:
        c_all = torch.empty(n_total, L, D, dtype=torch.float32, device=device)
        uc_row = torch.empty(1, L, D, dtype=torch.float32, device=device)
    dist.broadcast(c_all, src)
    dist.broadcast(uc_row, src)
    lo, hi = shard_range(n_total, world, rank)
    assert hi - lo == n_local, (lo, hi, n_local)
    return c_all[lo:hi].contiguous(), uc_row.expand(n_local, -1, -1).contiguous()


def gather_waveforms(wav_local, dist=None, dst=0, counts=None):
    """C2.  wav_local [n_local, T] -> [n_total, T] on `dst` (None elsewhere); ragged shards allowed.
    `counts` = per-rank row counts when known up front (skips the count exchange and its host synchronisation)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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
