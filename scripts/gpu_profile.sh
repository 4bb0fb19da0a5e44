#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of one bench batch + a plain bench run.
# Usage: bash scripts/gpu_profile.sh <tag>   -> gpurun_out/prof_<tag>/..., gpurun_out/bench_<tag>.json
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err
find gpurun_out/prof_$tag -type f | head -20
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -c 1500 gpurun_out/bench_$tag.json
