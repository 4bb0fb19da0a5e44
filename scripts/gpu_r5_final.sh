#!/bin/bash
# Round-5 validation on the GPU box: the whole -m gpu suite, smoke, then the profile + bench passes
# (scripts/gpu_profile.sh: rocprofv3 kernel stats, FETCH / WRITE / SQ counter passes, pmc_traffic.json, the default bench line)
# and the secondary workloads' profiles (which also feed the secondary rooflines' traffic into profiles/pmc_traffic.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=10 2>&1 | tail -28 | tee gpurun_out/r5_gpu_tests_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 | tee gpurun_out/r5_smoke.txt
bash scripts/gpu_profile_secondary.sh r5 bf16x3
bash scripts/gpu_profile.sh r5 bf16x3
cp gpurun_out/bench_detail.json gpurun_out/r5_bf16x3_bench_detail.json
