#!/bin/bash
# Round 4, call 4: the whole -m gpu suite on the current tree (row chains, DDIM signature, decode_spec, knob funnel), then the new
# secondary bench lines on their own.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 -p no:cacheprovider --durations=8 2>&1 | tail -24 | tee gpurun_out/r4_call4_tests.txt
timeout 600 python bench.py --steps 3 --warmup 1 --secondary-only t2a_bf16,t2a_bigvgan,tool_latency > gpurun_out/r4_call4_bench_secondary.json 2> gpurun_out/r4_call4.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4_call4_bench_secondary.json'))
print('headline', d['value'], d['ms_per_step'], d.get('one_batch_in_flight'), d.get('one_batch_two_streams'))
print('box', d.get('box'))
for k, v in d.get('secondary', {}).items():
    print(k, {q: v.get(q) for q in ('value', 'ms_per_step', 'error', 'one_batch_in_flight', 'parity', 'T2A_txt2audio', 'I2A_img2audio')})
    print('   cpu', v.get('cpu_baseline'), (v.get('roofline') or {}).get('kernel'), (v.get('roofline') or {}).get('frac'))
PY
tail -5 gpurun_out/r4_call4.err
