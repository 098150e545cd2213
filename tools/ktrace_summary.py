"""Per-kernel mean durations from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
print("%-70s %7s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "share"))
for k in sorted(acc, key=lambda k: -sum(acc[k]))[:12]:
    v = acc[k]
    print("%-70s %7d %10.1f %10.1f %10.1f %6.1f%%" % (k[:70], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
