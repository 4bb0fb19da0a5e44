#!/bin/bash
# Round 2, fourth GPU call: 64x320 tile probes for the short-K linears, LDS-stage probes of the round-1 kernels on what is
# left to them, the new tests, the mixed workload, then the profile pass (rocprofv3 stats + live traffic + bench).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_dma2.py tests/test_gpu_tools.py -q --timeout 800 -k "3,2,0 or 3,3,0 or Inpaint_inference_files or identical" 2>&1 | tail -6
echo "== in-pipeline probes (one process)"
X=MAA_DMA2_N320; Y=MAA_DMA2_N640
timeout 600 python scripts/dma2_inpipe.py \
  "default=" \
  "x:64x320ns3 K<2048=$X=3,3,0,1,0,0,2047" \
  "x:64x320ns2 K<2048=$X=3,2,0,1,0,0,2047" \
  "x:64x320ns3 K=320=$X=3,3,0,1,0,0,320" \
  "y:64x320ns3 K<2048=$Y=3,3,0,1,0,0,2047" \
  "xy:64x320ns3=$X=3,3,0,1,0,0,2047 $Y=3,3,0,1,0,0,2047" \
  "qkv:t0ns2=MAA_DMA2_N960=0,2,0,1,0 MAA_DMA2_N1920=0,2,0,1,0" \
  "default again=" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2_dma2_inpipe3.txt
echo "== round-1 kernel LDS stages (separate processes: MAA_DMA_NS is read once)"
for ns in "" "2,3,2" "2,3,3" "2,2,2"; do
  echo "-- MAA_DMA_NS=$ns"; MAA_DMA_NS=$ns timeout 300 python scripts/dma2_inpipe.py "ns=" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2_dma2_inpipe3.txt
done
echo "== mixed workload (config 5)"; timeout 900 python bench.py --workload mixed --steps 2 --warmup 1 > gpurun_out/r2_bf16x3_bench_mixed.json 2> gpurun_out/bench_mixed.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r2_bf16x3_bench_mixed.json")); r = d["roofline"]
    print("MIXED", d["value"], d["ms_per_step"]); print({k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "launches")}); print(list(r["kernel_time_ms"].items())[:12])
except Exception as e:
    print("mixed failed", e); print(open("gpurun_out/bench_mixed.err").read()[-2500:])
PY
echo "== profile pass"; bash scripts/gpu_profile.sh r2 bf16x3
