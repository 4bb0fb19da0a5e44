#!/bin/bash
# Round 5, call 8: the fused MRF pair of the narrow HiFi-GAN stages (halo_pair_kernel) -- bit-identity / parity tests, then the same-call
# A/B on BASELINE configs[2] (HiFi-GAN 64 x 1024 frames) against the two launches per pair (MAA_NO_PAIR=1), kernel stats of both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_config3.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "pair or halo or vocoder or config3 or hifigan" 2>&1 | tail -8 | tee gpurun_out/r5_call8_tests_tail.txt
run() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --workload hifigan64 --no-cpu-baseline --steps 4 2> gpurun_out/r5_call8_$tag.err | tee gpurun_out/r5_call8_$tag.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print('$tag', d['value'], d['ms_per_step'], r.get('kernel'), r.get('frac'), r.get('whole_pass_frac'))"
  python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); kt=(d.get('roofline') or {}).get('kernel_time_ms') or {}; print('    total', round(sum(kt.values()),1), list(kt.items())[:8])"
}
run pair_a MAA_NO_PAIR=0
run two_launches_a MAA_NO_PAIR=1
run pair_b MAA_NO_PAIR=0
run two_launches_b MAA_NO_PAIR=1
for v in 0 1; do
  MAA_NO_PAIR=$v timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_np$v -o bench -- python bench.py --workload hifigan64 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r5_call8_prof_np$v.json 2> gpurun_out/r5_call8_prof_np$v.err
  python scripts/prof_summary.py gpurun_out/prof_np$v/bench_results.db > gpurun_out/r5_call8_hifigan64_kernel_stats_no_pair${v}.txt
  rm -rf gpurun_out/prof_np$v
  head -12 gpurun_out/r5_call8_hifigan64_kernel_stats_no_pair${v}.txt | cut -c1-160
done
