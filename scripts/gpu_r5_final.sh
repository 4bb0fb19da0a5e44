#!/bin/bash
# Round-5 validation on the GPU box, inside what is left of the round's GPU budget (every section checks the clock; the order is
# the order of importance):
#   1. the whole -m gpu suite (four xdist workers: most of a test's wall time is the CPU oracle; whatever fails there is re-run serially)
#   2. smoke
#   3. rocprofv3 kernel stats + FETCH / WRITE / SQ counter passes of configs[1] -> profiles/pmc_traffic.json (SKIP_BENCH: the line comes last)
#   4. the default `python bench.py` line (picks the fresh configs[1] traffic table up: hash-matched)
#   5. kernel stats + counter passes of configs[2] (hifigan64) and configs[4] (mixed) while time is left: their tables land in
#      profiles/pmc_traffic.json for the NEXT bench run (the driver's) -- this call's own line shows the secondaries' traffic as null
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); BUDGET=${BUDGET:-1020}
left() { echo $(( BUDGET - ($(date +%s) - T0) )); }
stamp() { echo "[r5_final] $1: $(( $(date +%s) - T0 )) s elapsed, $(left) s left" | tee -a gpurun_out/r5_final_timeline.txt; }
rm -f gpurun_out/parity.jsonl gpurun_out/r5_final_timeline.txt

stamp start
timeout 420 python -m pytest tests -m gpu -q -n 4 --dist load --timeout 300 -o cache_dir=/tmp/pytest_cache_r5 --durations=10 2>&1 | tail -30 | tee gpurun_out/r5_gpu_tests_tail.txt
stamp "suite (xdist)"
if ! grep -Eq "^[0-9]+ passed" gpurun_out/r5_gpu_tests_tail.txt || grep -Eq "failed|error" gpurun_out/r5_gpu_tests_tail.txt; then
  timeout 240 python -m pytest tests -m gpu -q --lf --timeout 200 -o cache_dir=/tmp/pytest_cache_r5 2>&1 | tail -30 | tee gpurun_out/r5_gpu_tests_serial_rerun_tail.txt
  stamp "serial re-run of the failures"
fi
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 | tee gpurun_out/r5_smoke.txt
cp gpurun_out/parity.jsonl gpurun_out/r5_parity.jsonl 2>/dev/null
stamp smoke

SKIP_BENCH=1 bash scripts/gpu_profile.sh r5 bf16x3
stamp "configs[1] kernel stats + counter passes"
( time python bench.py > gpurun_out/r5_bf16x3_bench.json 2> gpurun_out/bench_r5.err ) 2> gpurun_out/r5_bench_wall_time.txt; tail -3 gpurun_out/r5_bench_wall_time.txt
cp gpurun_out/bench_detail.json gpurun_out/r5_bf16x3_bench_detail.json
stamp "default bench line"
python -c "
import json; d=json.load(open('gpurun_out/r5_bf16x3_bench.json')); print('VALUE', d['value'], d['ms_per_step'], 'one batch', (d.get('one_batch_in_flight') or {}).get('value')); r=d['roofline']; print({k:r.get(k) for k in ('kernel','achieved','peak','frac','avg_launch_us','traffic','mfma_busy')}); print(d['cpu_baseline'])
for k, v in d.get('secondary', {}).items(): print(k, {q: v.get(q) for q in ('value', 'ms_per_step', 'error')}, (v.get('roofline') or {}).get('kernel'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'))
print('line bytes', len(open('gpurun_out/r5_bf16x3_bench.json').read()))"
if [ $(left) -gt 150 ]; then WORKLOADS=hifigan64 bash scripts/gpu_profile_secondary.sh r5 bf16x3; stamp "configs[2] profile"; fi
if [ $(left) -gt 260 ]; then WORKLOADS=mixed bash scripts/gpu_profile_secondary.sh r5 bf16x3; stamp "configs[4] profile"; fi
