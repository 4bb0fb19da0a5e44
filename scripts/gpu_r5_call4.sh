#!/bin/bash
# Round 5, call 4: ResBlock conv1 -> GroupNorm partials hand-over (bit-identity + parity tests), then same-call A/B: default,
# MAA_NO_PARTIALS=1, and a layout sweep of the one-pass GroupNorm (groups per workgroup / threads per workgroup).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_config5.py tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider -k "groupnorm or models or config" 2>&1 | tail -12 | tee gpurun_out/r5_call4_tests_tail.txt
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 6 2> gpurun_out/r5_call4_$tag.err | tee gpurun_out/r5_call4_$tag.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', 'inflight3', d['value'], 'one', d['one_batch_in_flight']['value'], 'other', d['one_batch_other_form']['value'], r['kernel'], r['frac'], r['avg_launch_us'])"
  python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); kt=d['roofline']['kernel_time_ms']; print('    groupnorm', kt.get('groupnorm'), 'reduce', {k: v for k, v in kt.items() if 'reduce' in k}, 'total', round(sum(kt.values()),1), 'top', list(kt.items())[:6])"
}
run default X=1
run no_partials MAA_NO_PARTIALS=1
run gpb2 MAA_GN_GPB=2
run gpb2_t512 MAA_GN_GPB=2 MAA_GN_THREADS=512
run gpb8 MAA_GN_GPB=8
run gpb1 MAA_GN_GPB=1
run default_again X=1
