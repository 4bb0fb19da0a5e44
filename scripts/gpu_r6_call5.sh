#!/bin/bash
# round 6 call 5: the whole GPU suite (serial, with per-test durations), then this round's profile passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=45 ) > gpurun_out/r6_call5_gpu_tests.txt 2>&1
tail -60 gpurun_out/r6_call5_gpu_tests.txt
bash scripts/gpu_profile.sh r6 bf16x3 > gpurun_out/r6_call5_profile.log 2>&1; tail -25 gpurun_out/r6_call5_profile.log
bash scripts/gpu_profile_secondary.sh r6 bf16x3 > gpurun_out/r6_call5_profile_secondary.log 2>&1; tail -30 gpurun_out/r6_call5_profile_secondary.log
