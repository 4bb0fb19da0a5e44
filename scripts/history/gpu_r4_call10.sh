#!/bin/bash
# Round 4, call 10: the row chains are now opt-in (MAA_ROWCHAIN=1).  Same-box A/B of default (launch per layer) against the
# chains with three batches in flight (the line also carries the one-batch and two-stream numbers and the new box.calib reads
# out of L2 / Infinity Cache), then the profile passes of the shipped default (pmc_traffic.json stamp + the bench line).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r4_call10_ab.txt; : > $out
run() { echo "## $*" >> $out; env $1 timeout 300 python bench.py --no-secondary --no-cpu-baseline ${@:2} 2>>gpurun_out/r4_call10.err | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    kt=(d.get('roofline') or {}).get('kernel_time_ms') or {}
    b=d.get('box') or {}
    print('value %.2f ms_per_step %.1f one %s two %s | eager ms/batch: total %.1f rowchain %.1f dma64 %.1f layernorm %.1f | sclk %s mclk %s fclk %s W %s calib %s' % (d['value'], d['ms_per_step'], (d.get('one_batch_in_flight') or {}).get('value'), (d.get('one_batch_two_streams') or {}).get('value'), sum(kt.values()), kt.get('igemm_rowchain_bf16x3<64x320>', 0.0), kt.get('igemm_dma_bf16x3<64x64>', 0.0), kt.get('layernorm', 0.0), b.get('sclk_mhz_median'), b.get('mclk_mhz_median'), b.get('fclk_mhz_median'), b.get('socket_power_w_median'), {k: round(v) for k, v in b.get('calib', {}).items() if k != 'note'}))
" >> $out; }
run X=0 --inflight 3 --steps 6 --warmup 1
run MAA_ROWCHAIN=1 --inflight 3 --steps 6 --warmup 1
run X=0 --inflight 3 --steps 6 --warmup 1
run MAA_ROWCHAIN=1 --inflight 3 --steps 6 --warmup 1
cat $out
tail -3 gpurun_out/r4_call10.err
bash scripts/gpu_profile.sh r4 bf16x3
bash scripts/gpu_profile_secondary.sh r4 bf16x3
