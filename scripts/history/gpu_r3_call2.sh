#!/bin/bash
# Round 3, second GPU call: the ping-pong engine -- operator tests, micro-benchmark against the second engine, the UNet /
# configs[1] parity tests with it in the pipeline, a short bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 900 python -m pytest tests/test_gpu_pp.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -25 | tee gpurun_out/r3_call2_pp_tests_tail.txt
timeout 600 python scripts/pp_bench.py bf16x3 2>&1 | tee gpurun_out/r3_pp_bench.txt
timeout 900 python -m pytest tests/test_gpu_config2.py "tests/test_gpu_models.py" -m gpu -q --timeout 600 -p no:cacheprovider -k "config2_batch8 or unet or ddim" 2>&1 | tail -15 | tee gpurun_out/r3_call2_model_tests_tail.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r3_call2_bench.json 2> gpurun_out/r3_call2_bench.err
MAA_PP=off timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/r3_call2_bench_ppoff.json 2>> gpurun_out/r3_call2_bench.err
python - <<'PY'
import json
for f in ("gpurun_out/r3_call2_bench.json", "gpurun_out/r3_call2_bench_ppoff.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", round(d["value"], 2), "one", d.get("one_batch_in_flight"))
        r = d.get("roofline")
        if r:
            print({k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "launches")})
            print(list(r["kernel_time_ms"].items())[:12])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 gpurun_out/r3_call2_bench.err
