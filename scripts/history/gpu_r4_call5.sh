#!/bin/bash
# Round 4, call 5: parity of the kernels touched since call 4 (skip conv on the LDS-DMA engine, rowchain prefetch / pair stores,
# flash attention's idle waves), then one batch in flight before / after in the same call (the previous build = AUDIOGPT_AMD_LIB).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests/test_gpu_rowchain.py tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_ops.py tests/test_gpu_precision.py -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r4_call5_tests.txt
out=gpurun_out/r4_call5_ab.txt; : > $out
run() { echo "## $*" >> $out; timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline "$@" 2>>gpurun_out/r4_call5.err | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('value %.2f ms_per_step %.1f one %s two %s calib %s' % (d['value'], d['ms_per_step'], (d.get('one_batch_in_flight') or {}).get('value'), (d.get('one_batch_two_streams') or {}).get('value'), (d.get('box') or {}).get('calib')))
" >> $out; }
run --inflight 1 --steps 3 --warmup 1
run --inflight 3 --steps 6 --warmup 1
run --inflight 1 --steps 3 --warmup 1
cat $out
timeout 300 python scripts/shape_profile.py 5 bf16x3 > gpurun_out/r4_shapes_bf16x3_eager_v3.txt 2>>gpurun_out/r4_call5.err
head -45 gpurun_out/r4_shapes_bf16x3_eager_v3.txt
tail -3 gpurun_out/r4_call5.err
