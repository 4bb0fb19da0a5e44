#!/bin/bash
# round 6 call 1: the new Inpaint tool test, then the cross-request batching A/B VERDICT r5 #3 asks for (same box, same call)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_tools.py -x -q -k "Inpaint" 2>&1 | tail -15 > gpurun_out/r6_call1_tests_tail.txt
cat gpurun_out/r6_call1_tests_tail.txt
out=gpurun_out/r6_call1_batching_ab.txt; : > $out
run() { # label, args
  python bench.py --no-secondary --no-cpu-baseline --no-roofline "${@:2}" > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "$1 FAILED" >> $out; tail -5 gpurun_out/_l.err >> $out; return; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/_l.json')); print('%-34s value %8.2f audio-s/s  ms_per_step %9.2f  steps %d' % (sys.argv[1], d['value'], d['ms_per_step'], d['steps']))" "$1" >> $out
}
run "8x3 inflight (headline)"       --steps 12 --warmup 3
run "8x1 lanes"                     --steps 6 --warmup 2 --inflight 1
run "16x1 lanes"                    --steps 4 --warmup 1 --inflight 1 --prompts-per-gpu 16
run "16x1 one stream"               --steps 4 --warmup 1 --inflight 1 --prompts-per-gpu 16 --cfg-split 0
run "24x1 lanes"                    --steps 3 --warmup 1 --inflight 1 --prompts-per-gpu 24
run "24x1 one stream"               --steps 3 --warmup 1 --inflight 1 --prompts-per-gpu 24 --cfg-split 0
run "16x2 inflight"                 --steps 6 --warmup 2 --inflight 2 --prompts-per-gpu 16
run "12x2 inflight"                 --steps 6 --warmup 2 --inflight 2 --prompts-per-gpu 12
run "32x1 lanes"                    --steps 3 --warmup 1 --inflight 1 --prompts-per-gpu 32
run "8x3 inflight (again)"          --steps 12 --warmup 3
cat $out
