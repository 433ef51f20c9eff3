"""Diagnostics (GPU): does the KMAX = 20 register-resident sweep kernel (4 waves per SIMD, 128 VGPRs) compute right iterates?
Runs the batch solver and one / several outer iterations at K = 19, 20 (and 18, 22 as controls) against the CPU oracle with
the library named by INSIDER_HIP_LIB.    python tools/k20_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insider_amd import api, workloads, _lib
from oracle import c_oracle
c_oracle.build()
print("library:", _lib.LIB_PATH, flush=True)

def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))

rng = np.random.default_rng(3)
for K in (18, 19, 20, 22):
    B = 64
    X = rng.standard_normal((B, 60, K))
    G = np.einsum("bmk,bml->bkl", X, X)
    q = np.einsum("bmk,bm->bk", X, rng.standard_normal((B, 60)) * 3)
    w0 = rng.standard_normal((B, K)) * 0.1
    got, sw = api.strong_coordinate_descent(None, None, w0, 2.0, 0.4, XtX=G, Xty=q, tol=1e-9, seed=5, it=1, return_sweeps=True)
    ref = None
    try:
        ref = np.array([np.asarray(c_oracle.strong_cd_cov(w0[b], 2.0, 0.4, G[b], q[b], tol=1e-9, seed=5, it=1)[0]) for b in range(B)])
    except Exception as e:
        print("oracle strong_cd_cov unavailable:", repr(e))
    if ref is None:
        # KKT certificate instead of the oracle call
        g = q - np.einsum("bkl,bl->bk", G, got)
        la, l2 = 2.0 * 0.4, 2.0 * 0.6
        viol = np.where(got != 0, np.abs(g - l2 * got - la * np.sign(got)), np.maximum(np.abs(g) - la, 0))
        print(f"batch K={K}: max KKT violation {viol.max():.3e} sweeps {sw.min()}..{sw.max()} finite {np.isfinite(got).all()}", flush=True)
    else:
        print(f"batch K={K}: rel err {rel(got, ref):.3e}", flush=True)
    for iters, opts in ((0, dict(cd_pass1=0)), (0, dict(cd_pass1=48)), (3, dict(cd_pass1=64, cd_pass_ratio=2, cd_cold_iters=9))):
        w = workloads.small(n=201, p=73, level_counts=(6, 10, 4), K=K, f=0.1, lam=0.3, alpha=0.1, seed=860198)
        rs = np.random.default_rng(830)
        A = [np.asfortranarray(rs.standard_normal(a.shape) * 1e-3) for a in w.A0]
        C = np.asfortranarray(rs.standard_normal(w.C0.shape) * 1e-3)
        ds = api.InsiderData(w.X, w.levels, w.M_train, w.M_test)
        for k, v in opts.items():
            ds.set_option(k, v)
        ref = c_oracle.optimize(w.X, w.levels, w.n_levels, A, C, w.M_train, w.M_test, w.lam, w.lam, w.alpha, tuning=1,
                                max_iter=iters, seed=830)
        try:
            got = ds.optimize([a.copy(order="F") for a in A], C.copy(order="F"), K, w.lam, w.lam, w.alpha, tuning=1,
                              max_iter=iters, global_tol=-1.0, seed=830)
            print(f"optimize K={K} iters={iters} {opts}: col rel err {rel(got['column_factor'], ref['column_factor']):.3e} "
                  f"loss {got['loss']:.6g} vs {ref['loss']:.6g} cap_hits {int(ds.info('cap_hits'))} max_gene_sweeps {int(ds.info('max_gene_sweeps'))}", flush=True)
        except Exception as e:
            print(f"optimize K={K} iters={iters} {opts}: FAILED {e!r}", flush=True)
        ds.close()
