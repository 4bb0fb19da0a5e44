#!/bin/bash
# Throughput A/B of engine policies with the bench's three batches in flight (the regime the headline number is quoted in):
# one bench process per variant (some knobs are read once per process).  Usage: bash scripts/inflight_ab.sh out.txt "NAME|ENV=v ENV=v" ...
out=$1; shift
: > $out
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  line=$(env $envs timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); o=d.get('one_batch_in_flight') or {}
    print('%8.2f audio-s/s %8.1f ms/batch | one in flight %8.2f %8.1f' % (d['value'], d['ms_per_step'], o.get('value',0), o.get('ms_per_step',0)))
except Exception as e: print('failed', e)")
  printf "%-26s %s   %s\n" "$name" "$line" "$envs" | tee -a $out
done
