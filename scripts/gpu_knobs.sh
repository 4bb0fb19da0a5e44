#!/bin/bash
# Compare tile-choice knobs on a short graph-mode bench.  Usage: bash scripts/gpu_knobs.sh <precision>
prec=${1:-bf16x3}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $1"; env $1 python bench.py --steps 2 --warmup 1 --ddim-steps 20 --precision $prec --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],2))"; }
run "MAA_CONC_EFF=1,1,1"
run "MAA_CONC_EFF=0.55,0.85,1.0"
run "MAA_CONC_EFF=0.4,0.75,1.0"
run "MAA_FORCE_CFG=0"
run "MAA_FORCE_CFG=1"
run "MAA_FORCE_CFG=2"
