#!/bin/bash
# Round 2, call 8: batched-load epilogues (all engines) + halo conv1d for the narrow vocoder stages.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== op + engine tests"; timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dma2.py tests/test_gpu_precision.py -q -x --timeout 1200 2>&1 | tail -8
echo "== in-pipeline"; timeout 300 python scripts/dma2_inpipe.py "default=" "r1 engines=MAA_DMA2=off" 2>&1 | grep -v amdgpu.ids
echo "== bench"; timeout 900 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_call8.json 2> gpurun_out/bench_call8.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r2_bench_call8.json")); r = d["roofline"]
    print("VALUE", d["value"], d["ms_per_step"], d.get("one_batch_in_flight")); print({k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "launches")})
    print(r["all_igemm"]); print(list(r["kernel_time_ms"].items())[:14])
    for k, v in d.get("secondary", {}).items():
        print(k, {q: v.get(q) for q in ("value", "ms_per_step", "error")}); rr = v.get("roofline")
        if rr: print("   ", rr["kernel"], round(rr["frac"], 4), list(rr["kernel_time_ms"].items())[:8], rr.get("whole_pass"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/bench_call8.err").read()[-2500:])
PY
echo "== hifigan64 without the halo kernel"; MAA_NO_HALO=1 timeout 300 python bench.py --workload hifigan64 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NO_HALO', round(d['value'],1), round(d['ms_per_step'],1))"
