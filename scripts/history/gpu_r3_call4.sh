#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_pp.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -4
timeout 600 python scripts/pp_ablate.py short 2>&1 | tee gpurun_out/r3_pp_ablate_v3.txt
bash scripts/pp_inpipe.sh 2>&1 | tee gpurun_out/r3_pp_inpipe_policy_ab.txt
