#!/bin/bash
# Round 2, third GPU call: the whole -m gpu suite on the new default policy, the full bench line (with the hifigan64
# secondary workload), and a few more in-pipeline policy probes (GEGLU / short-K through the wide-tile engine).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== new rows first (N1, N2, config 3)"; timeout 1200 python -m pytest tests/test_gpu_nsf.py tests/test_gpu_diffsinger.py tests/test_gpu_config3.py -q --timeout 900 2>&1 | tail -25
echo "== full gpu suite"; timeout 2400 python -m pytest tests -m gpu -q --timeout 1800 -x --deselect tests/test_gpu_nsf.py --deselect tests/test_gpu_diffsinger.py --deselect tests/test_gpu_config3.py 2>&1 | tail -15
echo "== bench"; timeout 900 python bench.py > gpurun_out/r2_bench_call3.json 2> gpurun_out/r2_bench_call3.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r2_bench_call3.json")); r = d["roofline"]
    print("VALUE", d["value"], d["ms_per_step"]); print({k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "launches")})
    print(r["all_igemm"]); print(list(r["kernel_time_ms"].items())[:16]); print(d.get("cpu_baseline"))
    s = d.get("secondary", {}).get("hifigan64", {})
    print("HIFIGAN64", {k: s.get(k) for k in ("value", "ms_per_step", "error")}); 
    if "roofline" in s:
        rr = s["roofline"]; print({k: rr[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "launches")}); print(rr["whole_pass"]); print(list(rr["kernel_time_ms"].items())[:12])
    print(s.get("cpu_baseline"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r2_bench_call3.err").read()[-2500:])
PY
echo "== in-pipeline probes"
Y=MAA_DMA2
timeout 600 python scripts/dma2_inpipe.py \
  "default=" \
  "r1=$Y=off" \
  "geglu:t0p4=MAA_DMA2_N2560=0,4,1,1,0 MAA_DMA2_N5120=0,4,1,1,0" \
  "geglu:t1p3=MAA_DMA2_N2560=1,3,1,1,0 MAA_DMA2_N5120=1,3,1,1,0" \
  "geglu:t0ns2=MAA_DMA2_N2560=0,2,0,1,0 MAA_DMA2_N5120=0,2,0,1,0" \
  "qkv:t0ns2=MAA_DMA2_N960=0,2,0,1,0 MAA_DMA2_N1920=0,2,0,1,0" \
  "k1280:t2S1=MAA_DMA2_N320=2,2,0,1,1024" \
  "default again=" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_dma2_inpipe2.txt
