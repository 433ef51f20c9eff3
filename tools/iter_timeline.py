"""Print the kernel timeline of one steady-state outer iteration from a rocprofv3 kernel trace (per queue)."""
import csv, glob, sys
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void insider::", "").replace("insider::", "")[:44], r["Queue_Id"]) for r in rows)
cd = [i for i, e in enumerate(ev) if e[2].startswith("k_cd_cols_reg")]
a, b = cd[-back], cd[-back + 1]
t0 = ev[a][1]
ints = sorted((e[0], e[1]) for e in ev[a + 1:b + 1])
cs, ce = ints[0]; tot = 0
for s, e in ints[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print("iteration span us %.1f  union busy us %.1f  kernels %d" % ((ev[b][1] - ev[a][1]) / 1e3, tot / 1e3, b - a))
qs = sorted(set(e[3] for e in ev[a + 1:b + 1]))
for e in ev[a + 1:b + 1]:
    print("%9.1f %9.1f %8.1f q%d %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, qs.index(e[3]), e[2]))
