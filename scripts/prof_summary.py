"""Summarise a rocprofv3 results database (rocprofv3 --kernel-trace --stats -d <dir> -o <name>) as text.

    python scripts/prof_summary.py gpurun_out/prof_r1a/bench_results.db > profiles/<round>_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats   (durations in ns as reported by rocprofv3; summary of %s)" % path)
    print("%-110s %9s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in rows:
        name = name.replace("maa::(anonymous namespace)::", "").replace("void ", "")
        if len(name) > 108:
            name = name[:105] + "..."
        print("%-110s %9d %14.1f %12.2f %6.2f%%" % (name, calls, total, avg, pct))
    try:
        vg = list(cur.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
                              "max(scratch_size) from kernels group by name order by sum(duration) desc limit 12"))
        print("\n# per-kernel resources (from the dispatch records)")
        print("%-90s %6s %6s %6s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch"))
        for name, v, a, s, l, sc in vg:
            name = name.replace("maa::(anonymous namespace)::", "").replace("void ", "")[:88]
            print("%-90s %6s %6s %6s %8s %8s" % (name, v, a, s, l, sc))
    except sqlite3.Error as e:
        print("# (no per-dispatch resource table: %s)" % e)


if __name__ == "__main__":
    main(sys.argv[1])
