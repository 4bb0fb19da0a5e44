#!/bin/bash
# gn_stats with eight loads in flight per thread (+ the flash-attention softmax changes) against the committed build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r3_call12_tests_tail.txt
bash scripts/lib_ab.sh audiogpt_amd/libaudiogpt_mi355x_prev.so groupnorm flash_attention 2>&1 | tee gpurun_out/r3_gn_flash_ab_v2.txt
