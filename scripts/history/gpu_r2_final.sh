#!/bin/bash
# Round-end validation on the GPU box: the whole -m gpu suite, then the profile + bench pass (scripts/gpu_profile.sh).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/r2_gpu_tests_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
bash scripts/gpu_profile.sh r2 bf16x3
