"""Merge the rocprofv3 --pmc passes of tools/pmc_issue.sh into one per-kernel table of issue-slot counters and the ratios
derived from them, and record the sweep / statistics kernels' figures in profiles/issue.json (read by bench.py, valid for
the library hash they were taken on).      usage: pmc_issue.py WORKLOAD OUT_PREFIX PASS_DIR...

Units (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_INST_CYCLES_* count
quad-cycles (4 shader cycles) summed over the waves / SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the
dispatch kept the XCD busy (collected in every pass, so every ratio is formed inside ONE pass); rocprofv3's counter CSV
carries each dispatch's start / end timestamps.  Per kernel, over all its launches of the command:
  clock_GHz       = GRBM_GUI_ACTIVE / 8 / duration                               the clock the part held in the kernel
  valu_busy       = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   share of SIMD time a VALU instruction executes
  salu_busy       = 4 SQ_INST_CYCLES_SALU / (same)
  waves_per_simd  = 4 SQ_WAVE_CYCLES / (same)                                    mean resident waves per SIMD
  valu_per_wave   = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES     share of a resident wave's time spent in VALU instructions
  wait_inst       = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        ... stalled at issue (dependency, port taken)
  wait_any        = SQ_WAIT_ANY / SQ_WAVE_CYCLES             ... parked (s_waitcnt, instruction fetch after a jump, barrier)
  cyc_per_valu    = 4 SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
  lanes           = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU                  mean active lanes per VALU instruction
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 x GRBM_GUI_ACTIVE / 8)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wl, out = sys.argv[1], sys.argv[2]
NSIMD, NCU, NXCD = 1024, 256, 8

# per pass: kernel -> {counter: sum, "_ns": summed duration, "_calls": n}
passes = []
for d in sys.argv[3:]:
    agg = collections.OrderedDict()
    seen = set()
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("insider::", "")
            if "k_" not in k:
                continue
            a = agg.setdefault(k, collections.defaultdict(float))
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"]))
                a["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                a["_calls"] += 1
    passes.append(agg)


def ratio(a, b, scale=1.0):
    return scale * a / b if (a is not None and b) else None


table = collections.OrderedDict()
for k in passes[0]:
    d = {"calls": int(passes[0][k]["_calls"]), "ms_per_call": passes[0][k]["_ns"] / passes[0][k]["_calls"] / 1e6, "counters": {}}
    der = {}
    for a in (p.get(k, {}) for p in passes):
        if not a:
            continue
        gui = a.get("GRBM_GUI_ACTIVE")
        simd = NSIMD * gui / NXCD if gui else None
        g = a.get
        for c, v in a.items():
            if not c.startswith("_") and c != "GRBM_GUI_ACTIVE":
                d["counters"][c] = v
        cand = {
            "clock_GHz": ratio(gui, a["_ns"], 1.0 / NXCD) if gui else None,
            "valu_busy": ratio(g("SQ_ACTIVE_INST_VALU"), simd, 4.0),
            "salu_busy": ratio(g("SQ_INST_CYCLES_SALU"), simd, 4.0),
            "waves_per_simd": ratio(g("SQ_WAVE_CYCLES"), simd, 4.0),
            "valu_per_wave": ratio(g("SQ_ACTIVE_INST_VALU"), g("SQ_WAVE_CYCLES")),
            "salu_per_wave": ratio(g("SQ_INST_CYCLES_SALU"), g("SQ_WAVE_CYCLES")),
            "wait_inst": ratio(g("SQ_WAIT_INST_ANY"), g("SQ_WAVE_CYCLES")),
            "cyc_per_valu": ratio(g("SQ_ACTIVE_INST_VALU"), g("SQ_INSTS_VALU"), 4.0),
            "salu_per_valu_inst": ratio(g("SQ_INSTS_SALU"), g("SQ_INSTS_VALU")),
            "mfma_busy": ratio(g("SQ_VALU_MFMA_BUSY_CYCLES"), NCU * 4 * gui / NXCD if gui else None),
            "mfma_f64_insts_per_call": ratio(g("SQ_INSTS_VALU_MFMA_F64"), a["_calls"]),
        }
        for name, v in cand.items():
            if v is not None and name not in der:
                der[name] = v
    # ratios whose two counters sit in different passes: per-wave shares against pass 1's SQ_WAVE_CYCLES (same command, same work)
    wc = passes[0][k].get("SQ_WAVE_CYCLES")
    av = passes[0][k].get("SQ_ACTIVE_INST_VALU")
    allc = d["counters"]
    der["wait_any"] = ratio(allc.get("SQ_WAIT_ANY"), wc)
    der["active_any"] = ratio(allc.get("SQ_ACTIVE_INST_ANY"), wc)
    der["lanes"] = ratio(allc.get("SQ_THREAD_CYCLES_VALU"), av)
    d["derived"] = der
    table[k] = d

json.dump(table, open(out + ".json", "w"), indent=1)
cols = ["clock_GHz", "valu_busy", "salu_busy", "waves_per_simd", "valu_per_wave", "salu_per_wave", "wait_inst", "wait_any", "active_any",
        "cyc_per_valu", "salu_per_valu_inst", "lanes", "mfma_busy", "mfma_f64_insts_per_call"]
with open(out + ".csv", "w") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "ms_per_call"] + cols)
    for k, d in sorted(table.items(), key=lambda kv: -(kv[1]["calls"] * kv[1]["ms_per_call"])):
        w.writerow([k, d["calls"], f"{d['ms_per_call']:.4f}"] + [("" if d["derived"].get(c) is None else f"{d['derived'][c]:.4g}") for c in cols])
    w.writerow([])
    names = sorted({c for d in table.values() for c in d["counters"]})
    w.writerow(["kernel (counter sums over all calls)"] + names)
    for k, d in table.items():
        w.writerow([k] + [f"{d['counters'].get(c, float('nan')):.6g}" for c in names])
print(open(out + ".csv").read().split("\n\n")[0])

# profiles/issue.json: the sweep and the statistics kernel of this workload, keyed by the hash of the library that ran
from insider_amd import _build  # noqa: E402
ipath = os.environ.get("INSIDER_ISSUE_JSON") or os.path.join(ROOT, "profiles", "issue.json")   # (on the GPU box: under gpurun_out/)
ij = json.load(open(ipath)) if os.path.exists(ipath) else {}
sha = _build.library_sha()
if ij.get("source_sha") != sha:
    ij = {}
# the GPU box has no .git: the commit comes from the job that starts the passes (INSIDER_COMMIT, set here before gpurun)
commit = os.environ.get("INSIDER_COMMIT", "").strip()
if not commit:
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
commit = commit or "not recorded (no .git on the GPU box; source_sha identifies the sources)"
extra = os.environ.get("INSIDER_PMC_EXTRA", "").strip()


def pick(pred):
    ks = [k for k in table if pred(k)]
    if not ks:
        return None
    k = max(ks, key=lambda k: table[k]["calls"] * table[k]["ms_per_call"])
    d = table[k]["derived"]
    return {"kernel": k, "launches": table[k]["calls"], "ms_per_launch": table[k]["ms_per_call"], "clock_GHz": d.get("clock_GHz"),
            "valu_busy_of_resident_simd_time": d.get("valu_busy"), "salu_busy": d.get("salu_busy"), "waves_per_simd": d.get("waves_per_simd"),
            "wave_time_shares": {"valu": d.get("valu_per_wave"), "salu": d.get("salu_per_wave"), "issue_stall": d.get("wait_inst"),
                                 "parked_waitcnt_or_fetch": d.get("wait_any")},
            "salu_per_valu_inst": d.get("salu_per_valu_inst"), "active_lanes_per_valu": d.get("lanes"), "mfma_busy": d.get("mfma_busy")}


# every entry carries ITS OWN command and commit (round 4's file had one top-level command, the last workload's)
base_wl = wl.split("_")[0]
ij["source_sha"] = sha
ij.pop("command", None)
ij.pop("commit", None)
ij[wl] = {"command": (f"rocprofv3 --pmc <SQ_* pass> -- python3 bench.py --workload {base_wl} --steps 4 --warmup 0 --no-cpu-baseline"
                      + (" " + extra if extra else "") + " (three passes, tools/pmc_issue.sh)"),
          "commit": commit,
          "sweep_kernel": pick(lambda k: k.startswith("k_cd_cols") and "false" not in k),
          "statistics_kernel": pick(lambda k: k.startswith(("k_col_paircnt", "k_col_factored", "k_list_stats")))}
json.dump(ij, open(ipath, "w"), indent=1)
print("profiles/issue.json updated for", wl, "source_sha", sha)
