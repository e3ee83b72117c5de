"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output) -> table on stdout and optional CSV.
usage: python dev/kstats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for name, calls, total, avg, mn, mx in rows:
        out.append((name, calls, int(total), round(avg, 1), round(100.0 * total / tot, 2), int(mn), int(mx)))
    for r in out[:40]:
        print(f"{str(r[0])[:90]:90s} {str(r[1]):>6s} {str(r[2]):>14s} {str(r[3]):>12s} {str(r[4]):>7s}")
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            csv.writer(f).writerows(out)


if __name__ == "__main__":
    main()
