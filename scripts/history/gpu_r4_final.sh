#!/bin/bash
# Round-end validation on the GPU box: the whole -m gpu suite, smoke, then the profile + bench passes
# (scripts/gpu_profile.sh: rocprofv3 kernel stats, FETCH / WRITE / SQ counter passes, pmc_traffic.json, the default bench line)
# and the secondary workloads' profiles (which also feed the secondary rooflines' traffic into profiles/pmc_traffic.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=10 2>&1 | tail -28 | tee gpurun_out/r4_gpu_tests_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
bash scripts/gpu_profile_secondary.sh r4 bf16x3
bash scripts/gpu_profile.sh r4 bf16x3
