#!/bin/bash
# Round 4, call 8: after dropping the GroupNorm partials and restoring the row-chain kernel of call 3 (v2): parity incl. batch
# invariance, then row chains on / off at one and three batches in flight, same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests/test_gpu_rowchain.py tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_config5.py tests/test_gpu_pp.py -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r4_call8_tests.txt
out=gpurun_out/r4_call8_ab.txt; : > $out
run() { echo "## $*" >> $out; env $1 timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline ${@:2} 2>>gpurun_out/r4_call8.err | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('value %.2f ms_per_step %.1f one %s calib %s' % (d['value'], d['ms_per_step'], (d.get('one_batch_in_flight') or {}).get('value'), {k: round(v) for k, v in (d.get('box') or {}).get('calib', {}).items() if k != 'note'}))
" >> $out; }
run X=0 --inflight 1 --steps 3 --warmup 1
run X=1 --inflight 1 --steps 3 --warmup 1
run MAA_ROWCHAIN=0 --inflight 1 --steps 3 --warmup 1
run MAA_ROWCHAIN=0 --inflight 1 --steps 3 --warmup 1
run X=0 --inflight 1 --steps 3 --warmup 1
run X=0 --inflight 3 --steps 6 --warmup 1
run MAA_ROWCHAIN=0 --inflight 3 --steps 6 --warmup 1
cat $out
tail -3 gpurun_out/r4_call8.err
