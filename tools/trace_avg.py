"""Average duration of the dominant kernels over the TIMED optimize() call of a profiled bench.py run, from rocprofv3's
kernel trace.  A column solve is one k_cd_cols_reg launch, or (cold outer iterations) a chain of limited passes separated
by k_pass_scatter; each optimize() call starts with one evaluation-only launch of the same kernel, and the warm-up call
precedes the timed one: the timed call's solves are the last `steps` solves of the run.
    python tools/trace_avg.py DIR [steps]"""
import csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 31
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)   # the newest run in the directory
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
# (round 3: the loss evaluation is a kernel of its own, k_cd_cols_reg<.., false>: not a solve)
chain = [r for r in rows if ("k_cd_cols" in r["Kernel_Name"] and ", false>" not in r["Kernel_Name"]) or "k_pass_scatter" in r["Kernel_Name"]]
solves, cur = [], []
for i, r in enumerate(chain):
    cur.append(r)
    nxt = chain[i + 1]["Kernel_Name"] if i + 1 < len(chain) else ""
    if "k_cd_cols" in r["Kernel_Name"] and "k_pass_scatter" not in nxt:
        solves.append(cur)
        cur = []
st = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_col_paircnt", "k_col_factored"))] or \
     [r for r in rows if "k_list_stats" in r["Kernel_Name"]][0::2]
timed = solves[-steps:]
t_cd = [sum(dur(r) for r in s) for s in timed]            # kernels of the solve (passes + scatters)
t_span = [(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e6 for s in timed]
t_st = [dur(r) for r in st[-steps:]]
print(f"column solves: {len(solves)} in the run ({sum(len(s) for s in solves)} launches); the timed call's {steps} solves average "
      f"{sum(t_span)/len(t_span):.3f} ms from first pass to last (kernel time {sum(t_cd)/len(t_cd):.3f} ms); outer iterations 0-4: "
      + ", ".join(f"{t:.2f}" for t in t_span[:5]) + f" ms ({', '.join(str(sum(1 for r in s if 'k_cd_cols' in r['Kernel_Name'])) for s in timed[:5])} passes); "
      f"median {sorted(t_span)[len(t_span)//2]:.3f} ms")
print(f"column statistics: {len(st)} launches; the timed call's average {sum(t_st)/len(t_st):.3f} ms")
ev = [r for r in rows if "k_cd_cols" in r["Kernel_Name"] and ", false>" in r["Kernel_Name"] and int(r["Start_Timestamp"]) < int(timed[0][0]["Start_Timestamp"])]
t0 = int(ev[-1]["Start_Timestamp"]) if ev else int(timed[0][0]["Start_Timestamp"])
t1 = int(solves[-1][-1]["End_Timestamp"])
print(f"timed call, first evaluation launch to last solve: {(t1 - t0)/1e6:.2f} ms = {(t1 - t0)/1e6/steps:.3f} ms per outer iteration")
