"""Print a rocprofv3 --stats kernel summary (kernel_stats.csv) sorted by total time."""
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)   # the newest run in the directory
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f, "total GPU ms", tot / 1e6)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-72s calls %6s total_ms %9.3f avg_us %9.1f pct %5.1f" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                     float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
