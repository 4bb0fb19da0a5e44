#!/bin/bash
# Round 4, call 11: model-level parity with the shipped default (row chains opt-in): configs[1] at full size + batch invariance,
# the tool chains, the chain on/off agreement, smoke.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity.jsonl
timeout 420 python -m pytest tests/test_gpu_config2.py tests/test_gpu_rowchain.py tests/test_gpu_tools.py tests/test_gpu_models.py -m gpu -x -q --timeout 300 -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r4_call11_tests.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2 | tee -a gpurun_out/r4_call11_tests.txt
