"""dev: timeline of the LAST fit in a rocprofv3 --kernel-trace CSV: kernel, duration, idle gap in front of it.
usage: python dev/timeline.py <dir with *_kernel_trace.csv> [min idle us that separates fits = 300] [which fit from the end = 1]"""
import csv, glob, sys, re
d = sys.argv[1]
sep = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
groups, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if (b[0] - a[1]) / 1e3 > sep:
        groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
g = groups[-which]
t0 = g[0][0]
busy = 0
agg = {}
print(f"# {len(groups)} groups; showing group -{which} with {len(g)} kernels, span {(g[-1][1]-t0)/1e3:.1f} us")
prev_end = t0
for s, e, name in g:
    nm = re.sub(r"^void ", "", name)
    nm = re.sub(r"\(.*", "", nm)[:60]
    print(f"{(s-t0)/1e3:9.1f}  gap {(s-prev_end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {nm}")
    busy += e - s
    a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
print(f"# busy {busy/1e3:.1f} us of {(g[-1][1]-t0)/1e3:.1f} us")
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"#   {nm:60s} x{c:4d} {t:9.1f} us")
