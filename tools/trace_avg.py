"""Average duration of the dominant kernels over the TIMED optimize() call of a profiled bench.py run, from rocprofv3's
kernel trace: the last `steps` solve launches of k_cd_cols_reg (each optimize() call starts with one evaluation-only launch
of the same kernel, and the warm-up call precedes the timed one) and the statistics launches that belong to them.
    python tools/trace_avg.py DIR [steps]"""
import csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 31
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
cd = [r for r in rows if "k_cd_cols" in r["Kernel_Name"]]
st = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_col_paircnt", "k_col_factored"))] or \
     [r for r in rows if "k_list_stats" in r["Kernel_Name"]][0::2]
t_cd, t_st = [dur(r) for r in cd[-steps:]], [dur(r) for r in st[-steps:]]
print(f"k_cd_cols_reg: {len(cd)} launches in the run; the timed call's {steps} solves average {sum(t_cd)/len(t_cd):.3f} ms "
      f"(first three: {t_cd[0]:.2f}, {t_cd[1]:.2f}, {t_cd[2]:.2f}; median {sorted(t_cd)[len(t_cd)//2]:.3f})")
print(f"column statistics: {len(st)} launches; the timed call's average {sum(t_st)/len(t_st):.3f} ms")
t0, t1 = int(cd[-steps - 1]["Start_Timestamp"]), int(cd[-1]["End_Timestamp"])
print(f"timed call, first evaluation launch to last solve: {(t1 - t0)/1e6:.2f} ms = {(t1 - t0)/1e6/steps:.3f} ms per outer iteration")
