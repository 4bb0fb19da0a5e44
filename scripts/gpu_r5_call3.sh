#!/bin/bash
# Round 5, call 3: one-pass GroupNorm (norm.hip gn_fused_kernel) -- operator tests, model parity, then same-call A/B against the
# statistics + apply launches (MAA_GN_TWO_PASS=1).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_precision.py -m gpu -q --timeout 600 -p no:cacheprovider -k "not dma_engine_bit_identical" 2>&1 | tail -12 | tee gpurun_out/r5_call3_tests_tail.txt
for rep in 1 2; do
  for two in 0 1; do
    MAA_GN_TWO_PASS=$two timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 6 2> gpurun_out/r5_call3_ab_two${two}_$rep.err | tee gpurun_out/r5_call3_ab_two${two}_$rep.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('two_pass', $two, 'inflight3', d['value'], 'one', d['one_batch_in_flight']['value'], 'other', d['one_batch_other_form']['value'], r.get('top_kernel_share'), d['box'].get('class'))"
    python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); kt=d['roofline']['kernel_time_ms']; print('   groupnorm ms', kt.get('groupnorm'), 'total', sum(kt.values()))"
  done
done
