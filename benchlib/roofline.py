"""`roofline`: the dominant kernel's algorithmic rate against the MFMA peak, and the PMC attachments (traffic, mfma_busy)."""
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
from .common import MFMA_PER_FLOP, PEAK_TFLOPS      # noqa: E402


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
