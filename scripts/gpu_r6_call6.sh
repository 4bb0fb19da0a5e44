#!/bin/bash
# round 6 call 6: igemm tiles chosen by least total workgroup time (MAA_TILE_MODE=1) against least launch time (default), 3 batches in flight and 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
out=gpurun_out/r6_call6_tile_mode_ab.txt; : > $out
run() { # label, env, args
  env $2 python bench.py --no-secondary --no-cpu-baseline --no-roofline "${@:3}" > gpurun_out/_l.json 2> gpurun_out/_l.err || { echo "$1 FAILED" >> $out; tail -5 gpurun_out/_l.err >> $out; return; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/_l.json')); o=d.get('one_batch_in_flight') or {}; print('%-34s value %8.2f audio-s/s  ms_per_step %9.2f  one-batch %s  wav %s' % (sys.argv[1], d['value'], d['ms_per_step'], o.get('value'), d.get('wav_sha16')))" "$1" >> $out
}
run "8x3 default"            X=1 --steps 12 --warmup 3
run "8x3 tile mode 1"        MAA_TILE_MODE=1 --steps 12 --warmup 3
run "8x3 default"            X=1 --steps 12 --warmup 3
run "8x3 tile mode 1"        MAA_TILE_MODE=1 --steps 12 --warmup 3
cat $out
MAA_TILE_MODE=1 python bench.py --inflight 1 --cfg-split 0 --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --breakdown > gpurun_out/_b.json 2> gpurun_out/r6_call6_breakdown_tile_mode1.txt
grep -v "^\[bench\]" gpurun_out/r6_call6_breakdown_tile_mode1.txt | head -16
