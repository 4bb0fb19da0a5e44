#!/bin/bash
# round 6, call 22: the shared CFG prefix handed from the unconditional lane to the conditional lane (two-lane form): bit-identity
# tests, then the same-call A/B of one batch owning the GPU (lanes) and of the headline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_tools.py -x -q --durations=5 -k "guided_step or cfg_halves or config2 or serving_arrangement or T2A or pipeline_replica" ) > gpurun_out/r6_call22_tests.txt 2>&1
tail -12 gpurun_out/r6_call22_tests.txt
B="python bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 12 --warmup 3"
row() { python -c "
import json,sys
d=json.loads(open('gpurun_out/_b.json').read().strip().splitlines()[-1]); print('%-44s value %8.2f audio-s/s  ms_per_step %9.2f  one_batch %s' % (sys.argv[1], d['value'], d['ms_per_step'], (d.get('one_batch_in_flight') or {}).get('value')))" "$1"; }
: > gpurun_out/r6_call22_cfg_shared_lanes_ab.txt
for rep in 1 2; do
for sh in 1 0; do
  MAA_CFG_SHARED=$sh $B --inflight 1 --cfg-split 1 --steps 6 --no-one-batch > gpurun_out/_b.json 2> gpurun_out/_b.err || tail -5 gpurun_out/_b.err
  row "8x1 two lanes  MAA_CFG_SHARED=$sh (rep $rep)" >> gpurun_out/r6_call22_cfg_shared_lanes_ab.txt
done
done
for sh in 1 0; do
  MAA_CFG_SHARED=$sh $B > gpurun_out/_b.json 2> gpurun_out/_b.err || tail -5 gpurun_out/_b.err
  row "8x3 in flight  MAA_CFG_SHARED=$sh" >> gpurun_out/r6_call22_cfg_shared_lanes_ab.txt
done
cat gpurun_out/r6_call22_cfg_shared_lanes_ab.txt
