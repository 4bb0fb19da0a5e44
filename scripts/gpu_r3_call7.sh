#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_precision.py -m gpu -q --timeout 300 -p no:cacheprovider -k "attention" 2>&1 | tail -4
timeout 300 python scripts/attn_bench.py bf16x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_attn_bench_v2.txt
echo "== MAA_FLASH_OCC4=1" | tee -a gpurun_out/r3_attn_bench_v2.txt
MAA_FLASH_OCC4=1 timeout 300 python scripts/attn_bench.py bf16x3 2>&1 | grep -v amdgpu.ids | head -2 | tee -a gpurun_out/r3_attn_bench_v2.txt
