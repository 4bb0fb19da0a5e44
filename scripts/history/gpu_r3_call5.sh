#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_pp.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -6
timeout 600 python scripts/pp_bench.py bf16x3 2>&1 | head -22 | tee gpurun_out/r3_pp_bench_v4.txt
cat > /tmp/ab.sh <<'AB'
source scripts/pp_inpipe.sh
AB
python - <<'PY'
import re
s = open("scripts/pp_inpipe.sh").read()
head = s[:s.index('run "engines off')]
open("/tmp/ab2.sh", "w").write(head + 'run "engines off (round-2 kernels)" MAA_PP=off MAA_PP1=off\nrun "default policy" MAA_PPX=0\nrun "conv default / 1x1 off" MAA_PP1=off\nrun "conv 128,3 where 160 is default" MAA_PP=128,3\n')
PY
bash /tmp/ab2.sh 2>&1 | tee gpurun_out/r3_pp_inpipe_policy_ab_v4.txt
