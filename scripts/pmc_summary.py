"""Per-kernel sums of the PMC counters in a rocprofv3 results db.  python scripts/pmc_summary.py <db> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
cur = db.cursor()
try:
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
except sqlite3.Error as e:
    print("no counters_collection view:", e)
    sys.exit(0)
print("# columns:", cols)
name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
val = "value" if "value" in cols else [c for c in cols if "value" in c][0]
gx = "grid_size_x" if "grid_size_x" in cols else ("grid_size" if "grid_size" in cols else None)
q = "select %s, %s, %s count(*), sum(%s) from counters_collection group by %s, %s%s" % (
    name_col, cname, (gx + "," if gx else "0,"), val, name_col, cname, ("," + gx) if gx else "")
rows = list(cur.execute(q))
agg = {}
for k, c, g, n, v in rows:
    k = k.replace("maa::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    agg.setdefault((k, g), {})[c] = (n, v)
def key(item):
    d = item[1]
    for c in ("SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE"):
        if c in d:
            return -d[c][1]
    return 0
for (k, g), d in sorted(agg.items(), key=key)[:top]:
    n = max(v[0] for v in d.values())
    print("%-64s grid %-9s launches %5d  " % (k, g, n) + "  ".join("%s=%.4g" % (c, v[1]) for c, v in sorted(d.items())))
