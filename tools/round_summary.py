"""Key figures of a round's committed measurements (profiles/<round>/bench_*.json ...) as plain text, for DESIGN.md section 7 and
profiles/<round>/README.md: no number in those files is typed by hand.        python tools/round_summary.py r05"""
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", tag)


def load(name):
    p = os.path.join(D, name)
    return json.loads(open(p).readline()) if os.path.exists(p) else None


def line(name):
    d = load(name)
    if not d:
        return f"{name}: missing"
    r, mg, cd = d["roofline"], d["masked_gram"], d["cd_kernel"]
    ss = r.get("steady_state") or {}
    sp = ss.get("avg_launch_ms_parts") or {}
    s = (f"{name}: value {d['value']:.1f} it/s ({d['ms_per_step']:.3f} ms/step, steps {d['steps']} warmup {d['warmup']}) | roofline.frac {r['frac']:.3f} "
         f"(statistics {r['avg_launch_ms_parts']['statistics']:.3f} + sweeps {r['avg_launch_ms_parts']['sweeps']:.3f} ms; steady {ss.get('frac', 0):.3f}: "
         f"{sp.get('statistics', 0):.3f} + {sp.get('sweeps', 0):.3f}) traffic {r.get('traffic')} | masked_gram.frac {mg['frac']:.3f} "
         f"({mg['avg_launch_ms']:.3f} ms, {mg['achieved']:.1f} TF) hbm {mg.get('hbm_GBs_measured')} | cd {cd['coordinate_updates_per_s'] / 1e9:.1f} G updates/s, "
         f"{cd['sweeps_per_gene_per_iter']:.0f} sweeps/gene/iter, share {cd['share_of_wall']:.2f}, model frac {cd.get('valu_issue_model_frac')}, cap {cd['cap_hits']}, "
         f"longest {cd['max_gene_sweeps']} | sha {d.get('library_source_sha')}")
    cb = d.get("cpu_baseline")
    if cb:
        live = (cb.get("model_check_live") or {}).get("measured_over_model")
        s += f" | cpu {cb['value']:.5f} it/s on {cb['cores']} threads ({cb.get('cpu_model')}), live check {live}, ratio {d['value'] / cb['value']:.0f}"
        cov = (cb.get("settings") or {}).get("covariance_form_variant")
        if cov:
            s += f", covariance-form variant {cov['value']:.3f}"
    if r.get("measured_copy_GBs"):
        s += f" | copy {r['measured_copy_GBs']:.0f} GB/s"
    return s


for f in sorted(glob.glob(os.path.join(D, "bench_*.json"))):
    print(line(os.path.basename(f)))
for f in sorted(glob.glob(os.path.join(D, "grid_*.json"))):
    d = json.loads(open(f).readline())
    g, gc, gw = d.get("grid", {}), d.get("grid_concurrent", {}), d.get("grid_warm_start", {})
    print(f"{os.path.basename(f)}: grid {g.get('wall_s', 0):.2f} s; concurrent {gc.get('concurrent')}: {gc.get('wall_s', 0):.2f} s = {gc.get('speedup_vs_serial_grid', 0):.2f} x, "
          f"identical {gc.get('identical_to_serial_grid')}; warm start {gw.get('wall_s', 0):.2f} s, best point agrees {gw.get('best_point_agrees_with_cold')}, "
          f"max |d test rmse| {gw.get('max_abs_test_rmse_difference_vs_cold')}")
p = os.path.join(D, "scale_projection.json")
if os.path.exists(p):
    d = json.load(open(p))
    print("scale_projection: single", round(d["single_gpu"]["value_it_per_s"], 1), "it/s;",
          {N: (round(e["projection"]["assumed"]["value_it_per_s"], 1), round(e["projection"]["assumed"]["speedup_vs_single_gpu"], 2),
               round(e["zero_cost_exchange"]["speedup_vs_single_gpu"], 2)) for N, e in d["N"].items()}, "sha", d.get("library_source_sha"))
for name in ("issue.json", "traffic.json"):
    p = os.path.join(os.path.dirname(D), name)
    if os.path.exists(p):
        j = json.load(open(p))
        print(name, "source_sha", j.get("source_sha"), {k: (v.get("commit"), (v.get("sweep_kernel") or {}).get("valu_busy_of_resident_simd_time"),
                                                       (v.get("sweep_kernel") or {}).get("clock_GHz"), (v.get("sweep_kernel") or {}).get("waves_per_simd"),
                                                       (v.get("sweep_kernel") or {}).get("wave_time_shares"), (v.get("statistics_kernel") or {}).get("mfma_busy"))
                                                   if isinstance(v, dict) and "sweep_kernel" in v else (v.get("commit") if isinstance(v, dict) else v)
                                                   for k, v in j.items() if k != "source_sha"})
