#!/bin/bash
# round 6, call 19: the new coverage tests (resblock "2" generators, DDIM host hooks) + the touched files
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( time timeout 800 python -m pytest tests/test_gpu_models.py tests/test_gpu_precision.py tests/test_gpu_tools.py -x -q --durations=10 -k "resblock2 or vocoder or hifigan or bigvgan or ddim_sampler" ) > gpurun_out/r6_call19_tests.txt 2>&1
tail -25 gpurun_out/r6_call19_tests.txt
