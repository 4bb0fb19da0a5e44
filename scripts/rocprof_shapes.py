"""Group a rocprofv3 kernel trace by (kernel, grid) -> true per-shape kernel durations (no event overhead).
    python scripts/rocprof_shapes.py <results.db> [divisor]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, grid_x, grid_y, count(*), sum(duration), avg(duration), min(duration) from kernels "
                       "group by name, grid_x, grid_y order by sum(duration) desc"))
tot = sum(r[4] for r in rows)
print("total kernel time %.3f ms (/%g = %.3f ms)" % (tot / 1e6, div, tot / 1e6 / div))
for name, gx, gy, n, s, a, mn in rows[:60]:
    name = name.replace("maa::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-52s grid %8d x %4d  n %7.1f  avg %9.2f us  min %9.2f us  total %9.3f ms (%5.1f%%)" % (
        name[:52], gx // 256, gy, n / div, a / 1e3, mn / 1e3, s / 1e6 / div, 100.0 * s / tot))
