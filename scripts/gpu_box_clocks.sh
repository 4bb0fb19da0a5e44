#!/bin/bash
# What the box's GPU runs at while the benchmark runs: rocm-smi clocks / power sampled once a second during a short
# bench.py run (box-to-box spread of one binary is several percent; this records the box-side variables next to the number).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
log=gpurun_out/box_clocks_samples.csv; : > $log
echo "== idle"; rocm-smi --showclocks --showpower --showperflevel --showtemp 2>&1 | grep -v "^=\|^$" | head -30
timeout 400 python bench.py --steps 6 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline > gpurun_out/box_clocks_bench.json 2>/dev/null &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 >> $log
  sleep 1
done
wait $pid
echo "== bench"; python -c "
import json; d=json.loads(open('gpurun_out/box_clocks_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'one batch', d['one_batch_in_flight']['value'])"
echo "== samples under load (csv header + every 5th row)"
rocm-smi --showclocks --showpower --csv 2>/dev/null | head -1
awk 'NF && NR % 5 == 0' $log | head -40
