#!/bin/bash
# Round 4, call 13 (last GPU minutes): configs[4] parity, the engine bit-identity tests and the stale-override test on the shipped default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 170 python -m pytest tests/test_gpu_config5.py tests/test_gpu_precision.py tests/test_gpu_dma2.py -m gpu -x -q --timeout 160 -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r4_call13_tests.txt
