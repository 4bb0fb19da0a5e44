#!/bin/bash
# do the three replicas overlap better when their DDIM steps are out of phase?  (bench.py --stagger-ms, one box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for st in 0 2.6 1.3 0; do
  out=$(timeout 200 python bench.py --steps 6 --warmup 1 --stagger-ms $st --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
  python - "$st" "$out" <<'PY'
import json, sys
st, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print("stagger %4s ms: %7.2f audio-s/s  %6.1f ms per step   box %s" % (st, d["value"], d["ms_per_step"], {k: d["box"][k] for k in ("sclk_mhz_median", "socket_power_w_median")}), flush=True)
except Exception as e:
    print("stagger %s FAILED %s" % (st, line[-300:]), flush=True)
PY
done
