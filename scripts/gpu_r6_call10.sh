#!/bin/bash
# round 6 call 10: narrow output convolution with conflict-free weight reads: operator test, UNet parity, per-kernel time
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q -k "narrow_output" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "unet" 2>&1 | tail -3
python bench.py --inflight 1 --cfg-split 0 --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --breakdown > gpurun_out/_b.json 2> gpurun_out/r6_call10_breakdown.txt
grep -v "^\[bench\]" gpurun_out/r6_call10_breakdown.txt | grep "narrow\|up2\|pixel\|split32"
python bench.py --no-secondary --no-cpu-baseline --no-roofline --steps 12 --warmup 3 | cut -c1-300
