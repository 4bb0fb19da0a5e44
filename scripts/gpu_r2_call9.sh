#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== halo tests"; timeout 1200 python -m pytest tests/test_gpu_precision.py tests/test_gpu_config3.py tests/test_gpu_models.py -q --timeout 900 -k "halo or hifigan or bigvgan or vocoder or batch64" 2>&1 | tail -6
for e in "X=1" "MAA_HALO_TL64=256" "MAA_HALO_NO128=1" "MAA_NO_HALO=1"; do
  echo "-- $e"; env $e timeout 300 python bench.py --workload hifigan64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('HIFIGAN64', round(d['value'],1), round(d['ms_per_step'],1)); print('   ', list(d['roofline']['kernel_time_ms'].items())[:6])"
done
