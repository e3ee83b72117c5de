"""Per-kernel averages of the counters in rocprofv3 --pmc csv output directories.
usage: python dev/pmc_table.py gpurun_out/pmc_<tag> [...]"""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = (row["Kernel_Name"][:48], row["Counter_Name"])
                acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn, cn), (s, c) in sorted(acc.items()):
            if "petal" in kn: print(f"{d.split('/')[-1]:22s} {kn:50s} {cn:28s} {s / c:16.1f}  (n={c})")
