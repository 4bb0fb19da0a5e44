#!/bin/bash
# rocprofv3 kernel trace of a short bench (graph mode) for per-shape durations.  Usage: bash scripts/gpu_rocprof_short.sh <tag> <precision>
tag=${1:-x}; prec=${2:-bf16x3}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o short -- python bench.py --steps 1 --warmup 0 --ddim-steps 12 --precision $prec --no-cpu-baseline --no-roofline > gpurun_out/short_$tag.json 2> gpurun_out/short_$tag.err
python scripts/rocprof_shapes.py gpurun_out/prof_$tag/short_results.db 12 > gpurun_out/shapes_rocprof_$tag.txt
rm -f gpurun_out/prof_$tag/short_results.db
python bench.py --steps 2 --warmup 1 --precision $prec --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('VALUE', d['value'], d['ms_per_step']); print(json.dumps(d['roofline'])[:1200])"
head -50 gpurun_out/shapes_rocprof_$tag.txt
