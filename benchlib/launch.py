"""Process-level plumbing of bench.py: the CPU stand-ins of the --stub-cpu control-flow tests and the self-launcher for N > 1."""
import os
import sys

import torch

BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


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
           "--master-port", str(port), BENCH_PY] + list(argv)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)
