"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md) of `bench.py` into the
per-launch HBM byte figures of profiles/traffic.json, keyed to the sources they were measured on.

    python tools/pmc_traffic.py WORKLOAD FETCH_DIR WRITE_DIR [COMMIT] > merged into profiles/traffic.json

Per kernel the mean counter value over its launches is taken (KiB units); FETCH_SIZE is doubled (gfx950 tallies 128-B
requests at 64 B); bytes = 1024 x (2 FETCH + WRITE).  Groups: the column statistics (k_col_paircnt / k_col_factored /
k_list_stats column side + the k_mm_rows<.., false> product with the held-out level sums), the sweep kernel."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    sys.path.insert(0, ROOT)
    from insider_amd import _build
    return _build.source_sha()


def series(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("insider::", "")
            agg[k].append(float(r["Counter_Value"]))
    return agg


def means(d, counter):
    return {k: (sum(v) / len(v), len(v)) for k, v in series(d, counter).items()}


def main():
    name, fdir, wdir = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else "uncommitted"
    fe, wr = means(fdir, "FETCH_SIZE"), means(wdir, "WRITE_SIZE")
    per_kernel = {}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        per_kernel[k] = {"fetch_KiB": f, "write_KiB": w, "launches": max(nf, nw), "bytes_per_launch": 1024.0 * (2.0 * f + w)}

    def total(pred):
        return sum(v["bytes_per_launch"] for k, v in per_kernel.items() if pred(k))

    stats = total(lambda k: k.startswith(("k_col_paircnt", "k_col_factored")))
    if stats == 0:      # list path: the column side is the launch with p units; both sides share the kernel name
        stats = total(lambda k: k.startswith("k_list_stats")) / 2
    # the sweep kernel: a column solve is ONE launch from outer iteration 3 on (all genes: statistics records in, factors
    # out) and a chain of passes in the cold outer iterations 0-2 (the passes re-read the records of the genes still running
    # and save / restore 3 KP doubles per gene): the single-launch figure is the LAST launch of the run (outer iteration 3 of
    # the 4-step command), the largest launch and the mean over all launches are kept beside it
    fs, ws = series(fdir, "FETCH_SIZE"), series(wdir, "WRITE_SIZE")
    cdk = [k for k in fs if k.startswith("k_cd_cols") and not k.endswith("false>")]   # the solve launches (not the checkpoint evaluation)
    cd_launch = []
    for k in cdk:
        cd_launch += [1024.0 * (2.0 * f + w) for f, w in zip(fs[k], ws.get(k, [0.0] * len(fs[k])))]
    ent = {"col_stats_bytes_per_launch": stats + total(lambda k: k.startswith("k_mm_rows") and "false" in k) / 2,
           "cd_bytes_per_launch": cd_launch[-1] if cd_launch else 0.0,
           "cd_bytes_largest_pass": max(cd_launch) if cd_launch else 0.0,
           "cd_bytes_mean_over_launches": sum(cd_launch) / len(cd_launch) if cd_launch else 0.0,
           "per_kernel": {k: v for k, v in per_kernel.items() if v["bytes_per_launch"] > 1e6}}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(path)) if os.path.exists(path) else {}
    if tj.get("source_sha") != source_sha():
        tj = {}
    # every entry carries its own command and commit
    ent["command"] = (f"rocprofv3 --pmc FETCH_SIZE -- / --pmc WRITE_SIZE -- python3 bench.py --workload {name.split('_')[0]} --steps 4 "
                      f"--warmup 0 --no-cpu-baseline" + (" " + os.environ["INSIDER_PMC_EXTRA"] if os.environ.get("INSIDER_PMC_EXTRA") else "")
                      + " (separate passes)")
    ent["commit"] = os.environ.get("INSIDER_COMMIT", "").strip() or commit
    tj["source_sha"] = source_sha()
    tj.pop("command", None)
    tj.pop("commit", None)
    tj[name] = ent
    json.dump(tj, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in ent.items() if k != "per_kernel"}))


if __name__ == "__main__":
    main()
