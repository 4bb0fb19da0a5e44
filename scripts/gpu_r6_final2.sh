#!/bin/bash
# round 6 validation, second call (after the last source change: explicit-only concurrency hint): touched tests, profile passes re-stamped, default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_pp.py tests/test_gpu_config2.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -4 > gpurun_out/r6_final2_tests.txt; cat gpurun_out/r6_final2_tests.txt
bash scripts/gpu_profile.sh r6 bf16x3 > gpurun_out/r6_final_profile.log 2>&1; tail -22 gpurun_out/r6_final_profile.log
bash scripts/gpu_profile_secondary.sh r6 bf16x3 > gpurun_out/r6_final_profile_secondary.log 2>&1; tail -6 gpurun_out/r6_final_profile_secondary.log
python bench.py > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err; cut -c1-600 gpurun_out/r6_final_bench.json
