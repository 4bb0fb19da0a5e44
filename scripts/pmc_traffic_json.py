"""HBM-side bytes per launch of every implicit-GEMM kernel family, from the FETCH_SIZE / WRITE_SIZE passes of
scripts/gpu_profile.sh, stamped with what they were measured on so that bench.py can refuse stale numbers.

    python scripts/pmc_traffic_json.py <prof_pmc.txt> <precision> <ddim steps of the PMC workload> <out.json>
    python scripts/pmc_traffic_json.py <prof_pmc.txt> <precision> <units of the PMC workload> <out.json> --section hifigan64|mixed

With --section the table is merged into <out.json> under "secondary"[section] (bench.py's secondary workloads look their
dominant kernel up there); a unit is one generator pass (hifigan64) or one DDIM step of each tool (mixed).

FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived counters); on gfx950 FETCH_SIZE counts each 128-byte request as 64
bytes, so it is doubled (MI355X_MICROARCH.md, HBM section).  The two counters come from separate passes.  A bench label
maps to one or more rocprof kernel-name prefixes (a split-K contraction is its GEMM launch + its reduce launch).
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path, prec, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
section_name = sys.argv[sys.argv.index("--section") + 1] if "--section" in sys.argv else None
LABELS = {  # bench.py / maa_prof label -> (rocprof kernel prefixes, prefix whose launch count is the label's)
    "igemm_dma_bf16x3<64x64>": (["igemm_dma_kernel<64, 64"], "igemm_dma_kernel<64, 64"),
    "igemm_dma_bf16x3<128x64>": (["igemm_dma_kernel<128, 64"], "igemm_dma_kernel<128, 64"),
    "igemm_dma_bf16x3<128x128>": (["igemm_dma_kernel<128, 128"], "igemm_dma_kernel<128, 128"),
    "igemm_dma2_bf16x3<128x128,splitK>": (["igemm_dma2_kernel<128, 128", "splitk_reduce_kernel"], "igemm_dma2_kernel<128, 128"),
    "igemm_dma2_bf16x3<128x128>": (["igemm_dma2_kernel<128, 128"], "igemm_dma2_kernel<128, 128"),
    "igemm_pp_bf16x3<256x128,splitK>": (["igemm_pp_kernel<2, 2, 2, 2", "splitk_reduce_kernel"], "igemm_pp_kernel<2, 2, 2, 2"),
    "igemm_pp_bf16x3<256x160,splitK>": (["igemm_pp_kernel<1, 5, 4, 1", "splitk_reduce_kernel"], "igemm_pp_kernel<1, 5, 4, 1"),
    "igemm_pp_bf16x3<256x128>": (["igemm_pp_kernel<2, 2, 2, 2"], "igemm_pp_kernel<2, 2, 2, 2"),
    "igemm_pp_bf16x3<256x160>": (["igemm_pp_kernel<1, 5, 4, 1"], "igemm_pp_kernel<1, 5, 4, 1"),
    "igemm_pp1_bf16x3<256x128>": (["igemm_pp1_kernel<2, 2, 2, 2"], "igemm_pp1_kernel<2, 2, 2, 2"),
    "igemm_bf16x3<128x128>": (["igemm_bf16_kernel<128, 128, 2, 2, 3"], "igemm_bf16_kernel<128, 128, 2, 2, 3"),
    "igemm_bf16x3<128x64>": (["igemm_bf16_kernel<128, 64, 2, 2, 3"], "igemm_bf16_kernel<128, 64, 2, 2, 3"),
    "igemm_bf16x3<64x64>": (["igemm_bf16_kernel<64, 64, 2, 2, 3"], "igemm_bf16_kernel<64, 64, 2, 2, 3"),
    "igemm_bf16x3<256x32>": (["igemm_bf16_kernel<256, 32, 4, 1, 3"], "igemm_bf16_kernel<256, 32, 4, 1, 3"),
    "halo_conv1d_bf16x3<64>": (["halo_conv1d_kernel<64"], "halo_conv1d_kernel<64"),
    "halo_conv1d_bf16x3<32>": (["halo_conv1d_kernel<32"], "halo_conv1d_kernel<32"),
    "halo_pair_bf16x3<64>": (["halo_pair_kernel<64"], "halo_pair_kernel<64"),
    "halo_pair_bf16x3<32>": (["halo_pair_kernel<32"], "halo_pair_kernel<32"),
    "igemm_f32<64x64>": (["igemm_f32_kernel<64, 64"], "igemm_f32_kernel<64, 64"),
    "igemm_f32<128x64>": (["igemm_f32_kernel<128, 64"], "igemm_f32_kernel<128, 64"),
    "igemm_f32<128x128>": (["igemm_f32_kernel<128, 128"], "igemm_f32_kernel<128, 128"),
}
LABELS["igemm_pp_up2_bf16x3<256x160>"] = (["UP2:igemm_pp_kernel<1, 5, 4, 1"], "UP2:igemm_pp_kernel<1, 5, 4, 1")
LABELS["igemm_pp_up2_bf16x3<256x128>"] = (["UP2:igemm_pp_kernel<2, 2, 2, 2"], "UP2:igemm_pp_kernel<2, 2, 2, 2")


def match(name, prefix):
    """rocprof kernel name against a label's prefix; the UP2 instantiations of igemm_pp_kernel (last template argument true: the
    four-phase upsample convolution, round 6) are labels of their own ("UP2:" prefixes) and never count as the plain engine's."""
    up2 = name.startswith("igemm_pp_kernel<") and name.rstrip().rstrip("(").endswith(", true>")
    if prefix.startswith("UP2:"):
        return up2 and name.startswith(prefix[4:])
    return name.startswith(prefix) and not up2


rows = {"FETCH_SIZE": {}, "WRITE_SIZE": {}, "SQ_VALU_MFMA_BUSY_CYCLES": {}, "GRBM_GUI_ACTIVE": {}}   # counter -> kernel name -> (launches, sum)
section = None
for line in open(path):
    if line.startswith("=="):
        section = line.split()[1]
        continue
    m = re.search(r"^(.*?)\s+grid\s+\S+\s+launches\s+(\d+)", line)
    if m:
        # (a reduce launch shared by several split-K engines is attributed to each: an upper bound on their traffic)
        for c in ([section] if section in ("FETCH_SIZE", "WRITE_SIZE") else ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"] if section == "SQ_VALU_MFMA_BUSY_CYCLES" else []):
            v = re.search(c + r"=([0-9.eE+-]+)", line)
            if v:
                n0, s0 = rows[c].get(m.group(1).strip(), (0, 0.0))
                rows[c][m.group(1).strip()] = (n0 + int(m.group(2)), s0 + float(v.group(1)))
kernels = {}
for label, (prefixes, count_prefix) in LABELS.items():
    tot, n = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, 0
    for c in tot:
        for name, (ln, kb) in rows[c].items():
            if any(match(name, p) for p in prefixes):
                tot[c] += kb
            if c == "FETCH_SIZE" and match(name, count_prefix):
                n += ln
    if n:
        kernels[label] = {"hbm_bytes_per_launch": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / n,
                          "fetch_kb_sum": tot["FETCH_SIZE"], "write_kb_sum": tot["WRITE_SIZE"], "launches": n,
                          "launches_per_ddim_step": n / float(steps)}
        # matrix-pipe occupancy of the GEMM launch itself (the reduce launch has no MFMAs): SQ_VALU_MFMA_BUSY_CYCLES counts
        # busy cycles summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE the kernel's cycles summed over the 8 XCDs
        busy = sum(v for name, (ln, v) in rows["SQ_VALU_MFMA_BUSY_CYCLES"].items() if match(name, count_prefix))
        act = sum(v for name, (ln, v) in rows["GRBM_GUI_ACTIVE"].items() if match(name, count_prefix))
        if busy and act:
            kernels[label]["mfma_busy"] = busy / (act / 8.0 * 1024.0)
from audiogpt_amd.build import _source_hash  # noqa: E402
NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace passes over %s; HBM-side bytes = 2 x FETCH_SIZE (gfx950 "
        "counts 128-B requests as 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, per launch of the labelled contraction (a split-K "
        "contraction includes its reduce launch); Infinity-Cache hits are counted by these counters; mfma_busy = "
        "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) of the GEMM launch, from a third pass; source %s")
if section_name is None:
    doc = {"precision": prec, "source_hash": _source_hash(), "ddim_steps": steps, "kernels": kernels,
           "note": NOTE % ("%d eager DDIM steps of the benchmark batch (8 latents + CFG)" % steps, path)}
    if os.path.exists(out):          # keep the secondary workloads' tables of an earlier call on the same sources
        try:
            old = json.load(open(out))
            keep = {k: v for k, v in old.get("secondary", {}).items()
                    if v.get("source_hash") == doc["source_hash"] and v.get("precision") == prec}
            if keep:                 # each section carries its own stamp (the file's top-level one may be an older round's)
                doc["secondary"] = keep
        except Exception:
            pass
else:
    doc = json.load(open(out)) if os.path.exists(out) else {"precision": prec, "source_hash": _source_hash(), "kernels": {}}
    for v in kernels.values():
        v["launches_per_unit"] = v.pop("launches_per_ddim_step")
    doc.setdefault("secondary", {})[section_name] = {
        "precision": prec, "source_hash": _source_hash(), "units": steps, "kernels": kernels,
        "note": NOTE % ("%d unit(s) of `bench.py --workload %s` (a unit = one generator pass / one DDIM step of each tool)" % (steps, section_name), path)}
json.dump(doc, open(out, "w"), indent=1)
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["launches"]):
    print("%-40s %8.1f MB per launch  (%d launches, %.1f per unit)" % (k, v["hbm_bytes_per_launch"] / 1e6, v["launches"], v.get("launches_per_ddim_step", v.get("launches_per_unit", 0))))
