#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pp.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_config3.py tests/test_gpu_models.py -m gpu -q --timeout 600 -p no:cacheprovider -k "hifigan or vocoder or config3 or geglu or unet" 2>&1 | tail -6
timeout 300 python bench.py --workload hifigan64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3_hifigan64_pp_persistent.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_hifigan64_pp_persistent.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("hifigan64: %.1f audio-s/s  %.1f ms/step; dominant %s %.1f TFLOP/s frac %.3f" % (d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["frac"]))
print("   ", list(r["kernel_time_ms"].items())[:7])
PY
timeout 600 python scripts/pp_bench.py bf16x3 2>&1 | sed -n 10,12p | tee gpurun_out/r3_pp_bench_v5.txt
python - <<'PY'
s = open("scripts/pp_inpipe.sh").read()
head = s[:s.index('run "engines off')]
open("/tmp/ab3.sh", "w").write(head + 'run "default policy" MAA_PPX=0\n')
PY
bash /tmp/ab3.sh 2>&1 | tee gpurun_out/r3_pp_inpipe_v5.txt
