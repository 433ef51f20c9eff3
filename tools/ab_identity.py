"""A/B (GPU box): fits of one library build dumped for a bit-for-bit comparison with another build's.
    INSIDER_HIP_LIB=<build> python tools/ab_identity.py run <tag>     factors, sweep counts, trajectory -> gpurun_out/ab/<tag>.npz
    python tools/ab_identity.py cmp <tagA> <tagB>                     every array equal, bit for bit?
Cases: c2 in full (K = 20: the four-wave kernel), a 10000 x 8192 slab of c3 (K = 30), a K = 16 and a K = 32 fit, 31 outer
iterations each from the N(0, 1e-6) inits (cold multi-pass iterations included), and one deep fit (sub_tol 1e-7, 61 iterations)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "ab")


def run(tag):
    from insider_amd import api, workloads
    os.makedirs(OUT, exist_ok=True)
    out = {}
    cases = [("c2", dict(), 31, 1e-5), ("c3", dict(p=8192), 31, 1e-5), ("c3", dict(p=4096, K=16), 31, 1e-5),
             ("c3", dict(p=4096, K=32), 21, 1e-5), ("c2", dict(p=4096), 61, 1e-7)]
    for ci, (name, kw, iters, sub_tol) in enumerate(cases):
        w = workloads.make(name, **kw)
        ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
        A = [a.copy(order="F") for a in w.A0]
        C = w.C0.copy(order="F")
        res = ds.optimize(A, C, w.K, w.lam, w.lam, w.alpha, tuning=w.tuning, max_iter=iters - 1, global_tol=-1.0, seed=11,
                          sub_tol=sub_tol)
        sw = ds.sweeps()
        out[f"{ci}_C"] = C
        for i, a in enumerate(A):
            out[f"{ci}_A{i}"] = a
        out[f"{ci}_sweeps"] = np.asarray(sw)
        out[f"{ci}_traj"] = np.asarray(res["traj"])
        print(f"{tag} case {ci} {name} {kw}: K={w.K} iters={res['iters']} last-iteration sweeps mean {np.mean(sw):.1f} max {np.max(sw)}",
              flush=True)
        ds.close()
    np.savez(os.path.join(OUT, tag + ".npz"), **out)


def cmp(a, b):
    A, B = np.load(os.path.join(OUT, a + ".npz")), np.load(os.path.join(OUT, b + ".npz"))
    bad = 0
    for k in A.files:
        same = A[k].shape == B[k].shape and A[k].tobytes() == B[k].tobytes()
        if not same:
            bad += 1
            d = np.max(np.abs(A[k].astype(float) - B[k].astype(float))) if A[k].shape == B[k].shape else float("nan")
            print(f"DIFFERENT {k}: max abs diff {d}")
    print(f"ab_identity {a} vs {b}: {len(A.files)} arrays, {bad} different -> {'IDENTICAL' if bad == 0 else 'NOT IDENTICAL'}")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
