"""The ONE JSON line on stdout (<= 6 kB) and the full record beside it (gpurun_out/bench_detail.json)."""
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
