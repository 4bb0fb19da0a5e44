#!/bin/bash
# round 6 call 12: five-wave attention workgroups (160 query rows) where that takes fewer waves, against 128-row workgroups everywhere (MAA_FLASH5=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
out=gpurun_out/r6_call12_flash5_ab.txt; : > $out
run() { # label, env, args
  env $2 python bench.py --no-secondary --no-cpu-baseline --no-roofline "${@:3}" > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "$1 FAILED" >> $out; tail -5 gpurun_out/_l.err >> $out; return; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/_l.json')); o=d.get('one_batch_in_flight') or {}; print('%-44s value %8.2f audio-s/s  ms_per_step %9.2f  one-batch %s' % (sys.argv[1], d['value'], d['ms_per_step'], o.get('value')))" "$1" >> $out
}
run "8x3 five-wave attention workgroups"     X=1 --steps 12 --warmup 3
run "8x3 128-row workgroups (MAA_FLASH5=0)"  MAA_FLASH5=0 --steps 12 --warmup 3
run "8x3 five-wave attention workgroups"     X=1 --steps 12 --warmup 3
run "8x3 128-row workgroups (MAA_FLASH5=0)"  MAA_FLASH5=0 --steps 12 --warmup 3
cat $out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_precision.py -x -q -k "attention" 2>&1 | tail -3
