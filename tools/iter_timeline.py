"""Print the kernel timeline of one steady-state outer iteration from a rocprofv3 kernel trace (per queue).
An outer iteration starts with the row phase's first kernel (k_mm_rows<.., true>: V = C A') after a column solve and ends
with the last sweep-kernel launch of its column step.    python tools/iter_timeline.py DIR [iterations back from the end]"""
import csv, glob, sys
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 6
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)   # the newest run in the directory
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void insider::", "").replace("insider::", "")[:44], r["Queue_Id"]) for r in rows)
starts, seen_cd = [], True
for i, e in enumerate(ev):
    if e[2].startswith("k_cd_cols_reg"):
        seen_cd = True
    elif seen_cd and (e[2].startswith("k_wgemm<") or e[2].startswith("k_wsyrk<") or
                      (e[2].startswith("k_mm_rows") and ", true>" in e[2]) or e[2].startswith("k_mm_reduce")):
        # the row phase's first kernel: a level Gram GEMM, C'C or V = C A', whichever stream wins
        starts.append(i)
        seen_cd = False
a, b = starts[-back - 1], starts[-back]
it = ev[a:b]
t0 = it[0][0]
end = max(e[1] for e in it if e[2].startswith("k_cd_cols_reg"))
ints = sorted((e[0], e[1]) for e in it)
cs, ce = ints[0]; tot = 0
for s, e in ints[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print("iteration span us %.1f (first row-phase kernel to the end of the column solve)  next iteration starts at %.1f  union busy us %.1f  kernels %d"
      % ((end - t0) / 1e3, (ev[b][0] - t0) / 1e3, tot / 1e3, len(it)))
qs = sorted(set(e[3] for e in it))
for e in it:
    print("%9.1f %9.1f %8.1f q%d %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, qs.index(e[3]), e[2]))

# means over the steady-state iterations of the run's last call (all but its first five)
per = []
for k in range(len(starts) - 1):
    seg = ev[starts[k]:starts[k + 1]]
    st = [e for e in seg if e[2].startswith(("k_col_paircnt", "k_col_factored", "k_list_stats"))]
    cd = [e for e in seg if e[2].startswith("k_cd_cols_reg") and ", false>" not in e[2]]
    if not st or not cd:
        continue
    per.append(((st[-1][0] - seg[0][0]) / 1e3, (st[-1][1] - st[-1][0]) / 1e3, (cd[-1][1] - cd[0][0]) / 1e3,
                (ev[starts[k + 1]][0] - seg[0][0]) / 1e3, len(cd)))
steady = [x for x in per[-26:] if x[4] == 1]
if steady:
    m = [sum(x[i] for x in steady) / len(steady) for i in range(4)]
    print("steady-state means over %d iterations (us): row phase (to the start of the column statistics) %.1f, statistics %.1f, "
          "sweeps %.1f, iteration period %.1f" % (len(steady), m[0], m[1], m[2], m[3]))
