#!/bin/bash
# rocprofv3 kernel trace of the HEADLINE arrangement (three batches in flight, graph replay): per-kernel durations under overlap, and how busy
# the chip is over time (sum of kernel durations / wall time of the traced window).  Usage: bash scripts/gpu_inflight_trace.sh <tag>
tag=${1:-r6}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_inflight_$tag -o bench -- python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --no-one-batch > gpurun_out/${tag}_inflight3_bench_under_rocprof.json 2> gpurun_out/_if.err
python scripts/prof_summary.py gpurun_out/prof_inflight_$tag/bench_results.db > gpurun_out/${tag}_inflight3_kernel_stats.txt
python - <<PY >> gpurun_out/${tag}_inflight3_kernel_stats.txt
import sqlite3, glob
db = sqlite3.connect("gpurun_out/prof_inflight_$tag/bench_results.db")
cur = db.cursor()
tab = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
rows = list(cur.execute("select start, end from %s order by start" % tab))
# the timed region = the last 3/4 of the dispatches (1 warm-up step of 4); overlap = sum of durations / union of busy intervals
rows = rows[len(rows) // 4:]
tot = sum(e - s for s, e in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = rows[-1][1] - rows[0][0]
print("\n# timed window of the trace: %d dispatches, wall %.1f ms, some kernel running %.1f ms (%.1f %%), sum of kernel durations %.1f ms = %.2f kernels running on average" % (len(rows), wall / 1e6, busy / 1e6, 100.0 * busy / wall, tot / 1e6, tot / wall))
PY
rm -rf gpurun_out/prof_inflight_$tag
head -16 gpurun_out/${tag}_inflight3_kernel_stats.txt | cut -c1-170; tail -2 gpurun_out/${tag}_inflight3_kernel_stats.txt
