#!/bin/bash
# Same-box A/B of two builds of the library: bench.py (configs[1], 3 steps) with AUDIOGPT_AMD_LIB pointing at each, twice,
# interleaved; prints the headline, the one-batch number and the per-batch kernel time of the labels named in $LABELS.
# Usage: bash scripts/lib_ab.sh <prev.so> [label ...] > gpurun_out/...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
prev="$1"; shift
LABELS="${*:-groupnorm flash_attention layernorm}"
run() {
  tag="$1"; lib="$2"
  out=$(AUDIOGPT_AMD_LIB="$lib" timeout 400 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$tag" "$out" $LABELS <<'PY'
import json, sys
tag, line, labels = sys.argv[1], sys.argv[2], sys.argv[3:]
try:
    d = json.loads(line)
    o = d.get("one_batch_in_flight") or {}
    kt = (d.get("roofline") or {}).get("kernel_time_ms") or {}
    tot = sum(kt.values())
    print("%-10s in-flight %d: %7.2f audio-s/s   one batch: %7.2f (%.0f ms)   kernel ms/batch: total %.1f  %s" % (
        tag, d["config"]["batches_in_flight"], d["value"], o.get("value", 0), o.get("ms_per_step", 0), tot,
        "  ".join("%s %.1f" % (k, kt.get(k, float("nan"))) for k in labels)), flush=True)
except Exception as e:
    print("%-10s FAILED %s" % (tag, line[-300:]), flush=True)
PY
}
new="$PWD/audiogpt_amd/libaudiogpt_mi355x.so"
run previous "$PWD/$prev"
run new "$new"
run previous "$PWD/$prev"
run new "$new"
