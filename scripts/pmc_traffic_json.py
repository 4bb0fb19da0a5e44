"""HBM-side bytes per launch of one kernel family from the PMC summary written by scripts/gpu_profile.sh.

python scripts/pmc_traffic_json.py <prof_pmc.txt> <rocprof kernel-name prefix> <bench family label> <precision> <out.json>

FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3 derived counters); on gfx950 FETCH_SIZE counts each 128-byte request as
64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM section).  The two counters come from separate passes."""
import json
import re
import sys

path, prefix, family, prec, out = sys.argv[1:6]
sums = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
launches = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
section = None
for line in open(path):
    if line.startswith("=="):
        section = line.split()[1]
        continue
    if section in sums and line.startswith(prefix):
        m = re.search(r"launches\s+(\d+)", line)
        v = re.search(section + r"=([0-9.eE+-]+)", line)
        if m and v:
            sums[section] += float(v.group(1))
            launches[section] += int(m.group(1))
n = max(launches.values())
if n == 0:
    print("no rows match", prefix)
    sys.exit(1)
per_launch = (2.0 * sums["FETCH_SIZE"] / max(launches["FETCH_SIZE"], 1) + sums["WRITE_SIZE"] / max(launches["WRITE_SIZE"], 1)) * 1024.0
json.dump({"precision": prec, "kernel_family": family, "rocprof_kernel_prefix": prefix,
           "hbm_bytes_per_launch": per_launch, "fetch_kb_sum": sums["FETCH_SIZE"], "write_kb_sum": sums["WRITE_SIZE"],
           "launches": n,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace passes over 4 eager DDIM steps "
                   "(8 latents + CFG), all launches whose kernel name starts with '%s'; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); source " % prefix + path},
          open(out, "w"), indent=1)
print(family, "HBM-side bytes per launch: %.1f MB over %d launches" % (per_launch / 1e6, n))
