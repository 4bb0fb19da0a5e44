#!/bin/bash
# Round 4, call 6: the row-chain prefetch with unconditional loads behind the DMA prologue, the time-embedding rows hoisted out of the
# DDIM loop: parity, then one batch / three batches in flight.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests/test_gpu_rowchain.py tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_config5.py "tests/test_gpu_tools.py::test_ddim_sampler_and_model_surface_match_reference" "tests/test_gpu_tools.py::test_ddim_sampler_mask_eta_intermediates_match_reference" -m gpu -x -q --timeout 600 -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r4_call6_tests.txt
out=gpurun_out/r4_call6_ab.txt; : > $out
run() { echo "## $*" >> $out; timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline "$@" 2>>gpurun_out/r4_call6.err | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    print('value %.2f ms_per_step %.1f one %s two %s calib %s' % (d['value'], d['ms_per_step'], (d.get('one_batch_in_flight') or {}).get('value'), (d.get('one_batch_two_streams') or {}).get('value'), {k: round(v) for k, v in (d.get('box') or {}).get('calib', {}).items() if k != 'note'}))
" >> $out; }
run --inflight 1 --steps 3 --warmup 1
run --inflight 3 --steps 6 --warmup 1
MAA_ROWCHAIN=0 run --inflight 1 --steps 3 --warmup 1
cat $out
timeout 300 python scripts/shape_profile.py 5 bf16x3 > gpurun_out/r4_shapes_bf16x3_eager_v4.txt 2>>gpurun_out/r4_call6.err
grep -E "^== unet|rc M|bd2 M12480|flash|groupnorm|layernorm|bg" gpurun_out/r4_shapes_bf16x3_eager_v4.txt | head -24
tail -3 gpurun_out/r4_call6.err
