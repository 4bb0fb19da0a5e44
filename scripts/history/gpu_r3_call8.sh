#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pp.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_config3.py tests/test_gpu_nsf.py tests/test_gpu_models.py -m gpu -q --timeout 600 -p no:cacheprovider -k "hifigan or vocoder or config3 or nsf or halo" 2>&1 | tail -8
for pp in default off; do
  if [ $pp = off ]; then export MAA_PP=off; fi
  timeout 300 python bench.py --workload hifigan64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_hifigan64_pp_$pp.json 2>/dev/null
  python - $pp <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r3_hifigan64_pp_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("hifigan64 MAA_PP=%s: %.1f audio-s/s  %.1f ms/step; dominant %s %.1f TFLOP/s frac %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["frac"]))
print("   ", list(r["kernel_time_ms"].items())[:7])
PY
done
