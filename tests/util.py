"""Shared helpers for the GPU parity tests."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _t64(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def rel_err(a, b):
    """(max|a-b| / max|b|, mean|a-b|, max|a-b|)."""
    a, b = _t64(a), _t64(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = (a - b).abs()
    return float(d.max() / b.abs().max().clamp_min(1e-30)), float(d.mean()), float(d.max())


def record(name, **kw):
    """Append a parity record to gpurun_out/parity.jsonl (merged back from the GPU box)."""
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity.jsonl"), "a") as f:
        f.write(json.dumps(dict(name=name, **kw)) + "\n")


def check(name, got, ref, rtol_max, atol_mean=None):
    assert torch.isfinite(_t64(got)).all(), name + ": non-finite output"
    r, mean, mx = rel_err(got, ref)
    record(name, rel_max=r, abs_mean=mean, abs_max=mx, tol=rtol_max)
    assert r <= rtol_max, f"{name}: rel max err {r:.3e} > {rtol_max:.1e} (abs max {mx:.3e}, abs mean {mean:.3e})"
    if atol_mean is not None:
        assert mean <= atol_mean, f"{name}: mean abs err {mean:.3e} > {atol_mean:.1e}"
    return r


def free_port():
    """A TCP port that is free right now on 127.0.0.1 (process-group rendezvous of the multi-process tests)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
