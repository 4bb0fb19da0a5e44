"""RCCL exercised on the one GPU a test box has: `bench.py --force-collectives` initialises the "nccl" (= RCCL)
process group with ONE rank and sends C1 (dist.scatter of the conditioning blocks + dist.broadcast of the unconditional row),
C2 (dist.gather of the waveforms), the barriers, the all_gather / all_reduce of the timings and ranks_seen's all_gather_object
down their collective branches -- device tensors, three batches in flight on private streams with the record_stream /
wait_event choreography of the N > 1 path -- and the waveforms must equal the plain single-process run bit for bit.  It proves
that the library loads, that device-tensor scatter / gather work on this torch build and that the stream / event choreography does
not deadlock, so that the driver's first 8-GPU run is not the first RCCL call."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(tmp_path, name, extra, port):
    out = str(tmp_path / (name + ".json"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--ddim-steps", "5",
           "--no-cpu-baseline", "--no-roofline", "--no-secondary", "--no-one-batch", "--json-out", out] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 6144
    return json.load(open(out)), json.loads(lines[0])


_PLAIN = {}


@pytest.mark.parametrize("inflight", [3, 1])
def test_rccl_world_of_one_runs_the_collective_branches(tmp_path, inflight):
    from tests.util import free_port
    if "run" not in _PLAIN:      # the plain single-process run, once: every step generates the same prompts, whatever is in flight
        _PLAIN["run"] = _bench(tmp_path, "plain", ["--inflight", "1"], free_port())[0]
    plain = _PLAIN["run"]
    forced, line = _bench(tmp_path, "forced", ["--inflight", str(inflight), "--force-collectives"], free_port())
    assert forced["config"]["batches_in_flight"] == inflight
    assert forced["ranks_seen"]["n_distinct"] == 1 and forced["ranks_seen"]["ids"][0].startswith(("pci:", "uuid:", "cuda-index:"))
    assert len(forced["per_rank_value"]) == 1 and forced["per_rank_value"][0] > 0
    assert "ranks_seen" not in plain
    # the collectives really ran: their device time is recorded per step, and is not the no-op's
    assert forced["comm_ms_per_step"]["C1_broadcast"] > 0 and forced["comm_ms_per_step"]["C2_gather"] > 0
    # same prompts, same start codes -> the gathered waveforms are the plain run's, bit for bit
    assert forced["wav_sha16"] == plain["wav_sha16"]
    assert line["ranks_seen"]["n_distinct"] == 1 and line["value"] > 0
