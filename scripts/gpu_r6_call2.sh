#!/bin/bash
# round 6 call 2: K slices of the conv engine x batch size (MAA_PP_S = "S at 10x78, S at 5x39"), slice- vs tile-major items
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_tools.py -x -q -k "show_mel" 2>&1 | tail -5 > gpurun_out/r6_call2_tests_tail.txt
cat gpurun_out/r6_call2_tests_tail.txt
out=gpurun_out/r6_call2_kslices_ab.txt; : > $out
run() { # label, env, args
  env $2 python bench.py --no-secondary --no-cpu-baseline --no-roofline "${@:3}" > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "$1 FAILED" >> $out; tail -5 gpurun_out/_l.err >> $out; return; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/_l.json')); print('%-44s value %8.2f audio-s/s  ms_per_step %9.2f  steps %d' % (sys.argv[1], d['value'], d['ms_per_step'], d['steps']))" "$1" >> $out
}
run "8x3 S=2,4 slice-major (new default)"  X=1 --steps 12 --warmup 3
run "8x3 S=2,4 tile-major (round 5)"       MAA_PP_TILE_MAJOR=1 --steps 12 --warmup 3
run "8x3 S=1,2"                            MAA_PP_S=1,2 --steps 12 --warmup 3
run "8x3 S=1,4"                            MAA_PP_S=1,4 --steps 12 --warmup 3
run "8x1 S=2,4"                            X=1 --steps 6 --warmup 2 --inflight 1
run "8x1 S=1,2"                            MAA_PP_S=1,2 --steps 6 --warmup 2 --inflight 1
for n in 16 20 24 32; do
  run "${n}x1 S=2,4"                       X=1 --steps 3 --warmup 1 --inflight 1 --prompts-per-gpu $n
  run "${n}x1 S=1,2"                       MAA_PP_S=1,2 --steps 3 --warmup 1 --inflight 1 --prompts-per-gpu $n
  run "${n}x1 S=1,1"                       MAA_PP_S=1,1 --steps 3 --warmup 1 --inflight 1 --prompts-per-gpu $n
done
run "16x2 S=2,4"                           X=1 --steps 6 --warmup 2 --inflight 2 --prompts-per-gpu 16
run "16x2 S=1,2"                           MAA_PP_S=1,2 --steps 6 --warmup 2 --inflight 2 --prompts-per-gpu 16
run "20x2 S=1,2"                           MAA_PP_S=1,2 --steps 6 --warmup 2 --inflight 2 --prompts-per-gpu 20
run "8x3 S=2,4 slice-major (again)"        X=1 --steps 12 --warmup 3
cat $out
