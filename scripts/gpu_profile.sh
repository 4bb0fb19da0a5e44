#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of one bench batch, PMC traffic passes, then the plain bench
# whose JSON line picks the fresh traffic up.  Everything lands in gpurun_out/ (copy what is to be judged into profiles/).
# Usage: bash scripts/gpu_profile.sh <tag> [precision]
tag=${1:-r2}; prec=${2:-bf16x3}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 1 --warmup 1 --inflight 1 --cfg-split 0 --precision $prec --no-cpu-baseline --no-secondary > gpurun_out/${tag}_${prec}_bench_under_rocprof.json 2> gpurun_out/bench_prof_$tag.err
python scripts/prof_summary.py gpurun_out/prof_$tag/bench_results.db > gpurun_out/${tag}_${prec}_kernel_stats.txt
python scripts/rocprof_shapes.py gpurun_out/prof_$tag/bench_results.db 300 > gpurun_out/${tag}_${prec}_kernel_shapes.txt
rm -rf gpurun_out/prof_$tag
# HBM-side traffic of the kernels (separate PMC passes, kernel-trace only), short eager workload with the bench's launch mix
STEPS=4
rm -f gpurun_out/${tag}_${prec}_pmc_fetch_write.txt
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d_ -f1-2 | tr -d ' ')
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${tag}_$n -o pmc -- python scripts/pmc_workload.py $prec $STEPS > gpurun_out/pmc_${tag}_$n.log 2>&1
  echo "== $(echo $set | cut -d' ' -f1)  [$set]  ($STEPS DDIM steps, 8 latents + CFG, $prec)" >> gpurun_out/${tag}_${prec}_pmc_fetch_write.txt
  python scripts/pmc_summary.py gpurun_out/pmc_${tag}_$n/pmc_results.db 60 | grep -v "^# columns" >> gpurun_out/${tag}_${prec}_pmc_fetch_write.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_$n gpurun_out/pmc_${tag}_$n.log
done
python scripts/pmc_traffic_json.py gpurun_out/${tag}_${prec}_pmc_fetch_write.txt $prec $STEPS profiles/pmc_traffic.json && cp profiles/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
[ -n "$SKIP_BENCH" ] && exit 0      # (re-stamp only: the bench line of these kernels already exists)
( time python bench.py --precision $prec > gpurun_out/${tag}_${prec}_bench.json 2> gpurun_out/bench_$tag.err ) 2> gpurun_out/${tag}_bench_wall_time.txt; tail -3 gpurun_out/${tag}_bench_wall_time.txt
python -c "
import json; d=json.load(open('gpurun_out/${tag}_${prec}_bench.json')); print('VALUE', d['value'], d['ms_per_step']); r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','peak','frac','avg_launch_us','launches','traffic')}); print(d['cpu_baseline'])
for k, v in d.get('secondary', {}).items(): print(k, {q: v.get(q) for q in ('value', 'ms_per_step', 'error')}, (v.get('roofline') or {}).get('kernel'), (v.get('roofline') or {}).get('frac'))"
head -14 gpurun_out/${tag}_${prec}_kernel_stats.txt | cut -c1-175
