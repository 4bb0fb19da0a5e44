#!/bin/bash
# Round 5, call 5: XCD-contiguous work order of GroupNorm / LayerNorm / flash attention / split-K reduce + tile-major K-slice
# items -- tests, same-call A/B against plain blockIdx order (MAA_XCD_ALIGN=0), rocprofv3 kernel stats of one batch in both forms.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py tests/test_gpu_pp.py tests/test_gpu_config2.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r5_call5_tests_tail.txt
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 6 2> gpurun_out/r5_call5_$tag.err | tee gpurun_out/r5_call5_$tag.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', 'inflight3', d['value'], 'one', d['one_batch_in_flight']['value'], 'other', d['one_batch_other_form']['value'], r['kernel'], r['frac'], r['avg_launch_us'])"
  python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); kt=d['roofline']['kernel_time_ms']; print('    total', round(sum(kt.values()),1), 'top', list(kt.items())[:9])"
}
run align1_a MAA_XCD_ALIGN=1
run align0_a MAA_XCD_ALIGN=0
run align1_b MAA_XCD_ALIGN=1
run align0_b MAA_XCD_ALIGN=0
for al in 1 0; do
  MAA_XCD_ALIGN=$al rocprofv3 --kernel-trace --stats -d gpurun_out/prof_al$al -o bench -- python bench.py --steps 1 --warmup 1 --inflight 1 --cfg-split 0 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r5_call5_prof_al$al.json 2> gpurun_out/r5_call5_prof_al$al.err
  python scripts/prof_summary.py gpurun_out/prof_al$al/bench_results.db > gpurun_out/r5_call5_kernel_stats_align${al}.txt
  rm -rf gpurun_out/prof_al$al
  head -18 gpurun_out/r5_call5_kernel_stats_align${al}.txt | cut -c1-175
done
