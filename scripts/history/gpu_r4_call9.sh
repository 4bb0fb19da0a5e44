#!/bin/bash
# Round 4, call 9: GroupNorm statistics with vector loads (default) against the scalar-load kernel (MAA_GN_SCALAR=1), same box:
# parity first (operators, model parity incl. batch invariance, the stale-override refusal), then one batch in flight with the
# per-label kernel time of an eager batch, three in flight, and four in flight.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dma2.py tests/test_gpu_config2.py tests/test_gpu_models.py tests/test_gpu_precision.py -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r4_call9_tests.txt
out=gpurun_out/r4_call9_ab.txt; : > $out
run() { echo "## $*" >> $out; env $1 timeout 300 python bench.py --no-secondary --no-cpu-baseline ${@:2} 2>>gpurun_out/r4_call9.err | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    kt=(d.get('roofline') or {}).get('kernel_time_ms') or {}
    print('value %.2f ms_per_step %.1f one %s two %s | eager ms/batch: total %.1f groupnorm %.1f flash %.1f | calib %s' % (d['value'], d['ms_per_step'], (d.get('one_batch_in_flight') or {}).get('value'), (d.get('one_batch_two_streams') or {}).get('value'), sum(kt.values()), kt.get('groupnorm', float('nan')), kt.get('flash_attention', float('nan')), {k: round(v) for k, v in (d.get('box') or {}).get('calib', {}).items() if k != 'note'}))
" >> $out; }
run X=0 --inflight 1 --steps 3 --warmup 1
run MAA_GN_SCALAR=1 --inflight 1 --steps 3 --warmup 1
run X=0 --inflight 1 --steps 3 --warmup 1
run MAA_GN_SCALAR=1 --inflight 1 --steps 3 --warmup 1
run X=0 --inflight 3 --steps 6 --warmup 1
run MAA_GN_SCALAR=1 --inflight 3 --steps 6 --warmup 1
run X=0 --inflight 4 --steps 8 --warmup 1
cat $out
tail -3 gpurun_out/r4_call9.err
