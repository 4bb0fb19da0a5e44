#!/bin/bash
# round 6 call 7: the whole GPU suite after the knob pruning / concurrency hint / test-time cuts, then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r6_call7_gpu_tests.txt 2>&1
tail -45 gpurun_out/r6_call7_gpu_tests.txt
( time python bench.py > gpurun_out/r6_call7_bench.json 2> gpurun_out/r6_call7_bench.err ) 2> gpurun_out/r6_call7_bench_wall.txt
tail -3 gpurun_out/r6_call7_bench_wall.txt; cut -c1-1500 gpurun_out/r6_call7_bench.json
