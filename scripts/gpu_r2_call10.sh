#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bit-identity (halo / generic / untiled snake) + bigvgan goldens"; timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_models.py tests/test_gpu_config2.py -q --timeout 800 -k "halo or bigvgan" 2>&1 | tail -4
echo "== mixed"; timeout 600 python bench.py --workload mixed --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MIXED', round(d['value'],2), round(d['ms_per_step'],1)); print(list(d['roofline']['kernel_time_ms'].items())[:10])"
echo "== shapes"; timeout 600 python scripts/shape_profile.py 3 bf16x3 2>/dev/null > gpurun_out/r2_shapes_bf16x3.txt; grep -A14 "== vae\|== vocoder" gpurun_out/r2_shapes_bf16x3.txt | cut -c1-150; grep -A30 "== unet" gpurun_out/r2_shapes_bf16x3.txt | cut -c1-150
