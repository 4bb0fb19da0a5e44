#!/bin/bash
# Round 4, call 12: after moving the MAA_ROWCHAIN policy switch from rowchain_covers() to the UNet (the operator entry point
# maa_op_rowchain runs the engine whatever the switch says): the row-chain tests, then the PMC passes again so that
# profiles/pmc_traffic.json carries the hash of the shipped sources (kernels unchanged since call 10, whose bench line is kept).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_rowchain.py -m gpu -x -q --timeout 180 -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r4_call12_tests.txt
SKIP_BENCH=1 bash scripts/gpu_profile.sh r4 bf16x3
bash scripts/gpu_profile_secondary.sh r4 bf16x3 > gpurun_out/r4_call12_secondary.log 2>&1
tail -3 gpurun_out/r4_call12_secondary.log
