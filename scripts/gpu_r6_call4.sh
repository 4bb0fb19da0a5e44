#!/bin/bash
# round 6 call 4: tests touched by the knob pruning, then the conv grid-cap experiment (fewer persistent workgroups per conv launch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_models.py tests/test_gpu_dma2.py tests/test_gpu_tools.py -x -q -n 4 2>&1 | tail -8 > gpurun_out/r6_call4_tests_tail.txt
cat gpurun_out/r6_call4_tests_tail.txt
out=gpurun_out/r6_call4_grid_cap_ab.txt; : > $out
run() { # label, env, args
  env $2 python bench.py --no-secondary --no-cpu-baseline --no-roofline "${@:3}" > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "$1 FAILED" >> $out; tail -5 gpurun_out/_l.err >> $out; return; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/_l.json')); o=d.get('one_batch_in_flight') or {}; print('%-34s value %8.2f audio-s/s  ms_per_step %9.2f  one-batch %s' % (sys.argv[1], d['value'], d['ms_per_step'], o.get('value')))" "$1" >> $out
}
run "8x3 default"            X=1 --steps 12 --warmup 3
run "8x3 conv grid cap 128"  MAA_PP_GRID=128 --steps 12 --warmup 3
run "8x3 conv grid cap 104"  MAA_PP_GRID=104 --steps 12 --warmup 3
run "8x3 conv grid cap 70"   MAA_PP_GRID=70 --steps 12 --warmup 3
run "8x3 cap 104, S=1,2"     "MAA_PP_GRID=104 MAA_PP_S=1,2" --steps 12 --warmup 3
run "8x3 cap 52, S=1,2"      "MAA_PP_GRID=52 MAA_PP_S=1,2" --steps 12 --warmup 3
run "8x4 cap 104"            MAA_PP_GRID=104 --steps 12 --warmup 4 --inflight 4
run "8x4 default"            X=1 --steps 12 --warmup 4 --inflight 4
run "8x3 default (again)"    X=1 --steps 12 --warmup 3
cat $out
