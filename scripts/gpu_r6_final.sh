#!/bin/bash
# round 6 validation: the whole GPU suite (serial), smoke, this round's profile passes (kernel stats + PMC, configs[1] / [2] / [4]), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r6_final_gpu_tests.txt 2>&1
tail -30 gpurun_out/r6_final_gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > gpurun_out/r6_final_smoke.txt; cat gpurun_out/r6_final_smoke.txt
bash scripts/gpu_profile.sh r6 bf16x3 > gpurun_out/r6_final_profile.log 2>&1; tail -22 gpurun_out/r6_final_profile.log
bash scripts/gpu_profile_secondary.sh r6 bf16x3 > gpurun_out/r6_final_profile_secondary.log 2>&1; tail -12 gpurun_out/r6_final_profile_secondary.log
