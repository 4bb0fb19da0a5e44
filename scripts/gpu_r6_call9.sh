#!/bin/bash
# round 6 call 9: the narrow output convolution + the four-phase upsample convolution: operator tests, model parity, same-call A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q -k "upsample_conv or narrow_output" 2>&1 | tail -25 > gpurun_out/r6_call9_tests_ops.txt; cat gpurun_out/r6_call9_tests_ops.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_config5.py tests/test_gpu_bf16_engines.py -x -q 2>&1 | tail -12 > gpurun_out/r6_call9_tests_models.txt; cat gpurun_out/r6_call9_tests_models.txt
out=gpurun_out/r6_call9_up2_narrow_ab.txt; : > $out
run() { # label, env, args
  env $2 python bench.py --no-secondary --no-cpu-baseline --no-roofline "${@:3}" > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "$1 FAILED" >> $out; tail -5 gpurun_out/_l.err >> $out; return; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/_l.json')); o=d.get('one_batch_in_flight') or {}; print('%-44s value %8.2f audio-s/s  ms_per_step %9.2f  one-batch %s' % (sys.argv[1], d['value'], d['ms_per_step'], o.get('value')))" "$1" >> $out
}
run "8x3 phase upsample conv + narrow out conv"   X=1 --steps 12 --warmup 3
run "8x3 round-5 paths (MAA_UP2=0)"               MAA_UP2=0 --steps 12 --warmup 3
run "8x3 phase upsample conv + narrow out conv"   X=1 --steps 12 --warmup 3
run "8x3 round-5 paths (MAA_UP2=0)"               MAA_UP2=0 --steps 12 --warmup 3
cat $out
python bench.py --inflight 1 --cfg-split 0 --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --breakdown > gpurun_out/_b.json 2> gpurun_out/r6_call9_breakdown.txt
grep -v "^\[bench\]" gpurun_out/r6_call9_breakdown.txt | head -26
