#!/bin/bash
# Round 4, call 3: the row-chain engine with interleaved fillers / register-resident rows: parity again, then the same A/Bs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 900 python -m pytest tests/test_gpu_rowchain.py tests/test_gpu_models.py -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r4_call3_tests.txt
out=gpurun_out/r4_rowchain_ab_v2.txt; : > $out
run() { echo "## $*" >> $out; env "$1" timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline --inflight 1 --steps 3 --warmup 1 2>>gpurun_out/r4_call3.err | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('value %.2f ms_per_step %.1f box %s' % (d['value'], d['ms_per_step'], json.dumps(d['box'])))
" >> $out; }
run MAA_ROWCHAIN=1
run MAA_ROWCHAIN=0
run MAA_ROWCHAIN=1
echo "## inflight 3, chains on / off" >> $out
for rc in 1 0; do
MAA_ROWCHAIN=$rc timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline --inflight 3 --steps 6 --warmup 1 2>>gpurun_out/r4_call3.err | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('value %.2f ms_per_step %.1f one %s' % (d['value'], d['ms_per_step'], d.get('one_batch_in_flight')))
" >> $out
done
cat $out
timeout 300 python scripts/shape_profile.py 5 bf16x3 > gpurun_out/r4_shapes_bf16x3_eager_v2.txt 2>>gpurun_out/r4_call3.err
grep -E "^== unet|rc M|layernorm|bd2 M12480" gpurun_out/r4_shapes_bf16x3_eager_v2.txt
tail -3 gpurun_out/r4_call3.err
