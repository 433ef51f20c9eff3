"""Merge rocprofv3 --pmc counter_collection.csv files (one per counter pass) into the per-kernel summary committed
under profiles/: kernel, counter, calls, grid sizes of the first calls, counter value per call (KiB for *_SIZE)."""
import csv, glob, sys, collections
rows = []
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "insider" not in k:
        continue
    agg.setdefault((k, r["Counter_Name"]), []).append((r.get("Grid_Size", r.get("Grid_Size_X", "")), float(r["Counter_Value"])))
w = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL)
w.writerow(["kernel", "counter", "calls", "grid_sizes", "values_KiB_per_call"])
for (k, c), v in sorted(agg.items()):
    w.writerow([k, c, len(v), " ".join(str(g) for g, _ in v[:6]), " ".join(str(int(round(x))) for _, x in v[:6])])
