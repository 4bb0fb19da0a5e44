#!/bin/bash
# Round 5, call 2: the in-kernel split-K finish of igemm_pp -- parity / bit-identity tests, then same-call A/B against the reduce
# launch (MAA_PP_REDUCE=1) on the headline (three replicas, one stream each) and on one batch (lanes); rocprofv3 kernel stats of
# one batch in both forms.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pp.py tests/test_gpu_models.py tests/test_gpu_rccl.py tests/test_gpu_config2.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -12 | tee gpurun_out/r5_call2_tests_tail.txt
for rep in 1 2; do
  for red in 0 1; do
    MAA_PP_REDUCE=$red timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 6 2> gpurun_out/r5_call2_ab_red${red}_$rep.err | tee gpurun_out/r5_call2_ab_red${red}_$rep.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('reduce_launch', $red, 'inflight3', d['value'], 'one', d['one_batch_in_flight']['value'], 'other', d['one_batch_other_form']['value'], {k: r.get(k) for k in ('kernel','frac','avg_launch_us','launches')}, r.get('top_kernel_share'), d['box'].get('class'))"
  done
done
for red in 0 1; do
  MAA_PP_REDUCE=$red rocprofv3 --kernel-trace --stats -d gpurun_out/prof_red$red -o bench -- python bench.py --steps 1 --warmup 1 --inflight 1 --cfg-split 0 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r5_call2_prof_red$red.json 2> gpurun_out/r5_call2_prof_red$red.err
  python scripts/prof_summary.py gpurun_out/prof_red$red/bench_results.db > gpurun_out/r5_call2_kernel_stats_reduce${red}.txt
  rm -rf gpurun_out/prof_red$red
  head -12 gpurun_out/r5_call2_kernel_stats_reduce${red}.txt | cut -c1-180
done
# the mixed workload: image-to-audio's CFG lanes beside the inpainting pipeline -- on / off
for split in 1 0; do
  MAA_CFG_SPLIT=$split timeout 300 python bench.py --workload mixed --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mixed cfg_split', $split, d['value'], d['ms_per_step'])"
done
