"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (diagnostic)."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "insider"
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if pat in k:
        agg[(k[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:40s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g} last={v[-1]:.4g}")
