#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for fl in 1 2 3; do
  echo "== bench --inflight $fl"; timeout 600 python bench.py --steps 6 --warmup 1 --inflight $fl --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('VALUE', round(d['value'],2), round(d['ms_per_step'],1), d.get('one_batch_in_flight'))"
done
echo "== mixed"; timeout 600 python bench.py --workload mixed --steps 2 --warmup 1 --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MIXED', round(d['value'],2), round(d['ms_per_step'],1))"
