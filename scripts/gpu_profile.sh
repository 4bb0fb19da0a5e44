#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of one bench batch, PMC traffic pass, plain bench.
# Usage: bash scripts/gpu_profile.sh <tag> [precision]  -> gpurun_out/prof_<tag>_*.txt, gpurun_out/bench_<tag>.json
tag=${1:-r1}; prec=${2:-bf16x3}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 1 --warmup 1 --precision $prec --no-cpu-baseline > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err
python scripts/prof_summary.py gpurun_out/prof_$tag/bench_results.db > gpurun_out/prof_${tag}_kernel_stats.txt
python scripts/rocprof_shapes.py gpurun_out/prof_$tag/bench_results.db 200 > gpurun_out/prof_${tag}_shapes.txt
rm -rf gpurun_out/prof_$tag
# HBM-side traffic of the kernels (separate PMC passes, kernel-trace only), short eager workload
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d_ -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${tag}_$n -o pmc -- python scripts/pmc_workload.py $prec 4 > gpurun_out/pmc_${tag}_$n.log 2>&1
  echo "== $set  (4 DDIM steps, 8 latents + CFG, $prec)" >> gpurun_out/prof_${tag}_pmc.txt
  python scripts/pmc_summary.py gpurun_out/pmc_${tag}_$n/pmc_results.db 14 | grep -v "^# columns" >> gpurun_out/prof_${tag}_pmc.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_$n gpurun_out/pmc_${tag}_$n.log
done
# HBM-side bytes per launch of the dominant kernel family -> read by bench.py for roofline.traffic
python scripts/pmc_traffic_json.py gpurun_out/prof_${tag}_pmc.txt "igemm_dma_kernel<64, 64" igemm_dma_bf16x3 $prec profiles/r1_pmc_traffic.json && cp profiles/r1_pmc_traffic.json gpurun_out/pmc_traffic_$tag.json
python bench.py --steps 3 --warmup 1 --precision $prec > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('VALUE', d['value'], d['ms_per_step']); r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','peak','frac','avg_launch_us','launches','traffic')}); print(d['cpu_baseline'])"
head -12 gpurun_out/prof_${tag}_kernel_stats.txt | cut -c1-175
cat gpurun_out/prof_${tag}_pmc.txt | cut -c1-220
