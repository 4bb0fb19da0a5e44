#!/bin/bash
# Compare kernel knobs (env vars) on a short graph-mode bench.  Usage: bash scripts/gpu_knobs.sh <precision> "<ENV1>" "<ENV2>" ...
prec=${1:-bf16x3}; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for kv in "$@"; do
  echo "== $kv"
  env $kv python bench.py --steps 2 --warmup 1 --ddim-steps 20 --precision $prec --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],2))"
done
