#!/bin/bash
# Round 5, call 6: the plain-bf16 mode (BASELINE configs[1]'s literal dtype) on the shipped engines -- TERMS = 1 instantiations of
# igemm_dma / igemm_dma2 / igemm_pp / igemm_pp1 -- tests, then the same-call A/B against the register-staged engine that mode
# ran on until now (MAA_NO_DMA=1 routes every contraction there), and the bf16x3 default beside them.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bf16_engines.py tests/test_gpu_precision.py tests/test_gpu_pp.py tests/test_gpu_dma2.py tests/test_gpu_config3.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r5_call6_tests_tail.txt
run() {
  tag=$1; prec=$2; shift; shift
  env "$@" timeout 400 python bench.py --precision $prec --no-secondary --no-cpu-baseline --steps 6 2> gpurun_out/r5_call6_$tag.err | tee gpurun_out/r5_call6_$tag.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['dtype'], 'inflight3', d['value'], 'one', d['one_batch_in_flight']['value'], r['kernel'], r['frac'], r['avg_launch_us'])"
  python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); kt=d['roofline']['kernel_time_ms']; print('    total', round(sum(kt.values()),1), 'top', list(kt.items())[:10])"
  cp gpurun_out/bench_detail.json gpurun_out/r5_call6_${tag}_detail.json
}
run bf16_shipped_engines_a bf16 MAA_X=0
run bf16_register_engine_a bf16 MAA_NO_DMA=1
run bf16x3_default_a bf16x3 MAA_X=0
run bf16_shipped_engines_b bf16 MAA_X=0
run bf16_register_engine_b bf16 MAA_NO_DMA=1
