#!/bin/bash
# round 6 validation after the shared CFG prefix / resblock "2" / DDIM host hooks: the whole GPU suite (serial), smoke, the profile
# passes re-stamped on these sources (kernel stats + PMC, configs[1] / [2] / [4]) and the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r6_final5_gpu_tests.txt 2>&1
tail -22 gpurun_out/r6_final5_gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -2 > gpurun_out/r6_final5_smoke.txt; cat gpurun_out/r6_final5_smoke.txt
bash scripts/gpu_profile.sh r6 bf16x3 > gpurun_out/r6_final5_profile.log 2>&1; tail -22 gpurun_out/r6_final5_profile.log
bash scripts/gpu_profile_secondary.sh r6 bf16x3 > gpurun_out/r6_final5_profile_secondary.log 2>&1; tail -8 gpurun_out/r6_final5_profile_secondary.log
