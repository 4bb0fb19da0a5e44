#!/bin/bash
# Round 4, call 1: does ONE batch of 8 prompts finish sooner as sub-batches on concurrent streams?  (results are batch-invariant
# bit for bit, so a split is exact.)  Same total prompts per line; no secondary / roofline / cpu legs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r4_split_probe.txt; : > $out
run() { echo "## $*" >> $out; timeout 300 python bench.py --no-secondary --no-roofline --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('value %.2f ms_per_step %.1f one %s box %s' % (d['value'], d['ms_per_step'], d.get('one_batch_in_flight'), d.get('box')))
" >> $out; }
run --inflight 1 --prompts-per-gpu 8 --steps 3 --warmup 1
run --inflight 2 --prompts-per-gpu 4 --steps 6 --warmup 1
run --inflight 4 --prompts-per-gpu 2 --steps 12 --warmup 1
run --inflight 3 --prompts-per-gpu 8 --steps 6 --warmup 1
run --inflight 6 --prompts-per-gpu 4 --steps 12 --warmup 1
run --inflight 2 --prompts-per-gpu 8 --steps 4 --warmup 1
run --inflight 4 --prompts-per-gpu 4 --steps 8 --warmup 1
cat $out
