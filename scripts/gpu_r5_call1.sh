#!/bin/bash
# Round 5, call 1: the new GPU tests (RCCL world of one, CFG lanes), the lanes A/B with one and three batches in flight, and the
# default bench line end to end (<= 6 kB on stdout, full record in gpurun_out/bench_detail.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_models.py tests/test_gpu_config2.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r5_call1_tests_tail.txt
for split in 1 0; do
  for k in 1 2; do
    timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 6 --cfg-split $split 2> gpurun_out/r5_call1_ab_${split}_$k.err | tee gpurun_out/r5_call1_ab_split${split}_$k.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split', $split, 'inflight3', d['value'], 'one', d['one_batch_in_flight']['value'], 'other', d.get('one_batch_other_form'), d['box'].get('class'), d['box']['calib'])"
  done
done
( time python bench.py > gpurun_out/r5_call1_bench_line.json 2> gpurun_out/r5_call1_bench.err ) 2> gpurun_out/r5_call1_bench_wall.txt
wc -c gpurun_out/r5_call1_bench_line.json; tail -3 gpurun_out/r5_call1_bench_wall.txt
cp gpurun_out/bench_detail.json gpurun_out/r5_call1_bench_detail.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r5_call1_bench_line.json'))
print('VALUE', d['value'], 'one', d['one_batch_in_flight'], d.get('one_batch_other_form'))
print(d['roofline']); print(d['box'])
for k,v in d['secondary'].items(): print(k, {q: v.get(q) for q in ('value','ms_per_step','error')}, (v.get('roofline') or {}).get('kernel'), (v.get('roofline') or {}).get('frac'))
P
