#!/bin/bash
# batches in flight: 1 .. 5 on one box with the round-3 engines (bench.py, configs[1])
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for n in 3 2 4 5 3 4; do
  k=$((n * 2)); [ $k -lt 6 ] && k=6
  out=$(timeout 400 python bench.py --steps $k --warmup 1 --inflight $n --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
  python - "$n" "$k" "$out" <<'PY'
import json, sys
n, k, line = sys.argv[1], sys.argv[2], sys.argv[3]
try:
    d = json.loads(line)
    o = d.get("one_batch_in_flight") or {}
    print("in flight %s (steps %s): %7.2f audio-s/s  %6.1f ms per step   latency in flight %.0f ms   one batch %.2f" % (
        n, k, d["value"], d["ms_per_step"], (d.get("batch_latency_ms") or {}).get("in_flight", 0), o.get("value", 0)), flush=True)
except Exception as e:
    print("in flight %s FAILED %s" % (n, line[-300:]), flush=True)
PY
done
