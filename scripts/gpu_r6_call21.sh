#!/bin/bash
# round 6, call 21: the shared-prefix tests again (fixed inputs) + every file a guided trajectory runs through
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( time timeout 1000 python -m pytest tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_config5.py tests/test_gpu_precision.py tests/test_gpu_bf16_engines.py tests/test_gpu_tools.py -x -q --durations=8 ) > gpurun_out/r6_call21_tests.txt 2>&1
tail -16 gpurun_out/r6_call21_tests.txt
