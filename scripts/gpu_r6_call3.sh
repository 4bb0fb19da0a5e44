#!/bin/bash
# round 6 call 3: per-kernel-family time / rate of one eager batch at 8 and at 32 prompts (where does a large batch still lose?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for n in 8 32; do
  python bench.py --prompts-per-gpu $n --inflight 1 --cfg-split 0 --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --breakdown \
     > gpurun_out/r6_call3_b${n}.json 2> gpurun_out/r6_call3_breakdown_b${n}.txt
  grep -v "^\[bench\]" gpurun_out/r6_call3_breakdown_b${n}.txt | head -40
done
