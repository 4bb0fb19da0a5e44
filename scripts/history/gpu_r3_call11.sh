#!/bin/bash
# GroupNorm as coalesced chunk partials + finish in the apply kernel; softmax scale folded into Q, masks only on ragged tiles
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_precision.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r3_call11_tests_tail.txt
bash scripts/lib_ab.sh audiogpt_amd/libaudiogpt_mi355x_prev.so groupnorm flash_attention 2>&1 | tee gpurun_out/r3_gn_flash_ab.txt
