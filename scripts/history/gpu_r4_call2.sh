#!/bin/bash
# Round 4, call 2: the row-chain engine -- operator parity, the UNet with / without it, the models' goldens; then same-box A/Bs of
# one batch in flight: chains on / off, the deeper copy queue for low-occupancy 64x64 launches, and a per-shape profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 900 python -m pytest tests/test_gpu_rowchain.py tests/test_gpu_models.py tests/test_gpu_config2.py -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r4_call2_tests.txt
out=gpurun_out/r4_rowchain_ab.txt; : > $out
run() { echo "## $*" >> $out; env "$1" timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline --inflight 1 --steps 3 --warmup 1 2>>gpurun_out/r4_call2.err | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('value %.2f ms_per_step %.1f box %s %s' % (d['value'], d['ms_per_step'], d['box']['sclk_mhz_median'], d['box']['socket_power_w_median']))
" >> $out; }
run MAA_ROWCHAIN=1
run MAA_ROWCHAIN=0
run MAA_DMA_NS_LOW=4
run MAA_DMA_NS_LOW=3
run MAA_ROWCHAIN=1
echo "## inflight 3, chains on" >> $out
timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline --inflight 3 --steps 6 --warmup 1 2>>gpurun_out/r4_call2.err | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('value %.2f ms_per_step %.1f one %s' % (d['value'], d['ms_per_step'], d.get('one_batch_in_flight')))
" >> $out
cat $out
timeout 300 python scripts/shape_profile.py 5 bf16x3 > gpurun_out/r4_shapes_bf16x3_eager.txt 2>>gpurun_out/r4_call2.err
head -60 gpurun_out/r4_shapes_bf16x3_eager.txt
tail -5 gpurun_out/r4_call2.err
