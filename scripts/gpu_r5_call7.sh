#!/bin/bash
# Round 5, call 7: q / k / v handed to the attention kernel as split32 rows (projection epilogues write them; flash_attn SPLIT_IN) --
# tests, same-call A/B against fp32 rows split per tile inside the kernel (MAA_ATTN_SPLIT=0 = rounds 1-4), rocprofv3 kernel stats of both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_precision.py tests/test_gpu_models.py tests/test_gpu_config2.py tests/test_gpu_config5.py tests/test_gpu_tools.py tests/test_gpu_encoders.py -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r5_call7_tests_tail.txt
run() {
  tag=$1; shift
  env "$@" timeout 400 python bench.py --no-secondary --no-cpu-baseline --steps 6 2> gpurun_out/r5_call7_$tag.err | tee gpurun_out/r5_call7_$tag.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['dtype'], 'inflight3', d['value'], 'one', d['one_batch_in_flight']['value'], r['kernel'], r['frac'], r['avg_launch_us'])"
  python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); kt=d['roofline']['kernel_time_ms']; print('    total', round(sum(kt.values()),1), 'top', list(kt.items())[:10])"
}
run split_a MAA_ATTN_SPLIT=1
run fp32_a MAA_ATTN_SPLIT=0
run split_b MAA_ATTN_SPLIT=1
run fp32_b MAA_ATTN_SPLIT=0
for v in 1 0; do
  MAA_ATTN_SPLIT=$v rocprofv3 --kernel-trace --stats -d gpurun_out/prof_as$v -o bench -- python bench.py --steps 1 --warmup 1 --inflight 1 --cfg-split 0 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r5_call7_prof_as$v.json 2> gpurun_out/r5_call7_prof_as$v.err
  python scripts/prof_summary.py gpurun_out/prof_as$v/bench_results.db > gpurun_out/r5_call7_kernel_stats_attn_split${v}.txt
  rm -rf gpurun_out/prof_as$v
  head -18 gpurun_out/r5_call7_kernel_stats_attn_split${v}.txt | cut -c1-175
done
