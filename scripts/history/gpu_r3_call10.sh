#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_precision.py tests/test_gpu_pp.py tests/test_gpu_dma2.py -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -6
python - <<'PY'
s = open("scripts/pp_inpipe.sh").read()
head = s[:s.index('run "engines off')]
open("/tmp/ab4.sh", "w").write(head + 'run "pair stores (default)" MAA_PPX=0\nrun "MAA_NO_PAIR_STORE=1" MAA_NO_PAIR_STORE=1\nrun "pair stores (default) again" MAA_PPX=0\n')
PY
bash /tmp/ab4.sh 2>&1 | tee gpurun_out/r3_pair_store_ab.txt
