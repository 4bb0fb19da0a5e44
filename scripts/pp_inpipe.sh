#!/bin/bash
# In-pipeline A/B of the ping-pong engines' policies: bench.py (configs[1], 3 steps) under each environment; prints the
# headline (batches in flight) and the one-batch number.  Usage: bash scripts/pp_inpipe.sh > gpurun_out/...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
  tag="$1"; shift
  out=$(env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline $EXTRA 2>/dev/null | tail -1)
  python - "$tag" "$out" <<'PY'
import json, sys
tag, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    o = d.get("one_batch_in_flight") or {}
    print("%-44s in-flight %d: %7.2f audio-s/s   one batch: %7.2f (%.0f ms)" % (tag, d["config"]["batches_in_flight"], d["value"], o.get("value", 0), o.get("ms_per_step", 0)), flush=True)
except Exception as e:
    print("%-44s FAILED %s" % (tag, line[-200:]), flush=True)
PY
}
run "engines off (round-2 kernels)"            MAA_PP=off MAA_PP1=off
run "default policy"                           MAA_PPX=0
run "conv pp 128,1 / 1x1 off"                  MAA_PP=128,1 MAA_PP1=off
run "conv pp 128,2 / 1x1 off"                  MAA_PP=128,2 MAA_PP1=off
run "conv pp 128,3 / 1x1 off"                  MAA_PP=128,3 MAA_PP1=off
run "conv default / 1x1 pp1 128,1 everywhere"  MAA_PP1=128,1
run "conv pp 128,1 / 1x1 pp1 128,1"            MAA_PP=128,1 MAA_PP1=128,1
EXTRA="--inflight 4" run "conv pp 128,1 / 1x1 pp1 128,1, 4 in flight" MAA_PP=128,1 MAA_PP1=128,1
EXTRA="--inflight 4" run "engines off, 4 in flight" MAA_PP=off MAA_PP1=off
