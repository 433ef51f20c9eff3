"""Independent numpy restatement of the INSIDER hot path — TEST INFRASTRUCTURE ONLY.

Written separately from insider_oracle.c, statement by statement after the
reference's Armadillo code, and used to cross-check the C oracle on small
cases (pure-Python loops: keep n*p below ~1e5).  PARITY UNPINNED (the
reference has no golden vectors; see insider_oracle.c's header).

Citations are to /root/reference (kai0511/insider).
"""
import numpy as np

M32 = 0xFFFFFFFF


# -- include/insider_perm.h restated in Python ints ----------------------------
def h32(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


PERM_PERIOD = 16384      # include/insider_perm.h: the order sequence of a solve repeats after this many sweeps


def perm_base(seed, it, sweep):
    sweep &= PERM_PERIOD - 1
    b = h32((seed & M32) ^ 0x9E3779B9)
    b = h32(b ^ ((seed >> 32) & M32) ^ ((0x85EBCA6B * it) & M32))
    b = h32((b + 0xC2B2AE35 * sweep) & M32)
    return b


def perm_key(base, l):
    return (h32(base ^ ((0x27D4EB2F * (l + 1)) & M32)) & 0xFFFFFFC0) | l


def sweep_order(inc, seed, unit, it, sweep, order_mode):
    if order_mode != 0:
        return list(inc)
    base = perm_base(seed, it, sweep)
    return sorted(inc, key=lambda l: perm_key(base, int(l)))


# -- src/utils.cpp:46-49 --------------------------------------------------------
def compute_sub_loss(residual, beta, lam, alpha):
    return np.sum(residual ** 2) / 2 + (1 - alpha) * lam * np.sum(beta ** 2) / 2 + alpha * lam * np.sum(np.abs(beta))


# -- src/coordinate_descent.cpp:56-127 ------------------------------------------
def strong_coordinate_descent(X, y, wstart, lam, alpha, XtX, Xty, tol=1e-5, seed=0, unit=0, it=0, order_mode=0,
                              max_sweeps=10000):
    X = np.asarray(X, float)
    y = np.asarray(y, float)
    K = X.shape[1]
    beta = np.array(wstart, float).copy()
    active = np.ones(K, bool)
    ex = np.abs(Xty) < alpha * (2 * lam - np.max(np.abs(Xty)))      # :74
    active[ex] = False                                               # :75
    beta[ex] = 0.0                                                   # :78
    residual = y - X @ beta                                          # :79
    iter_loss = compute_sub_loss(residual, beta, lam, alpha)         # :80
    sweep = 0
    while True:
        inc = np.flatnonzero(active)                                 # :83
        exi = np.flatnonzero(~active)                                # :84
        while True:
            pre_loss = iter_loss                                     # :87
            order = sweep_order(inc, seed, unit, it, sweep, order_mode)   # :89 (randperm -> deterministic)
            sweep += 1
            for k in order:                                          # :91-110
                u = residual @ X[:, k] + beta[k] * XtX[k, k]
                if abs(u) > lam * alpha:
                    upd = np.sign(u) * max(abs(u) - lam * alpha, 0.0) / (XtX[k, k] + lam * (1 - alpha))
                else:
                    upd = 0.0
                if upd != beta[k]:
                    residual = residual - (upd - beta[k]) * X[:, k]
                    beta[k] = upd
            iter_loss = compute_sub_loss(residual, beta, lam, alpha)  # :112
            if sweep >= max_sweeps or not abs(pre_loss - iter_loss) > tol:   # :114
                break
        if sweep >= max_sweeps:
            break
        grad = XtX[np.ix_(exi, inc)] @ beta[inc] - Xty[exi]          # :118
        viol = np.abs(grad) > alpha * lam                            # :119
        if not viol.any():
            break
        active[exi[viol]] = True                                     # :123
    return beta, sweep


def solve_sympd(A, b):
    return np.linalg.solve(A, b)


# -- src/optimize.cpp:139-198 ---------------------------------------------------
def optimize_row(residual, indicator, updating_factor, c_factor, updating_confd, gram, lam, tuning):
    out = updating_factor.copy()
    seq = np.unique(updating_confd)                                  # :147
    K = c_factor.shape[0]
    if tuning == 1:
        for lv in seq:
            XtX = np.zeros((K, K))
            Xty = np.zeros(K)
            for r in np.flatnonzero(updating_confd == lv):           # :159-161
                nz = np.flatnonzero(indicator[r, :])                 # :162
                zero = np.flatnonzero(indicator[r, :] == 0)          # :163
                outcome = residual[r, nz]                            # :166-167
                XtX += gram - c_factor[:, zero] @ c_factor[:, zero].T   # :170
                Xty += c_factor[:, nz] @ outcome                     # :171
            XtX[np.diag_indices(K)] += lam                           # :174
            out[lv - 1, :] = solve_sympd(XtX, Xty)                   # :175
    elif tuning == 0:
        Xtys = c_factor @ residual.T                                 # :180
        for lv in seq:
            ids = np.flatnonzero(updating_confd == lv)
            XtX = len(ids) * gram                                    # :186
            XtX[np.diag_indices(K)] += lam
            out[lv - 1, :] = solve_sympd(XtX, Xtys[:, ids].sum(axis=1))   # :188-190
    else:
        raise ValueError("Parameter tuning should be either 0 or 1!")
    return out


# -- src/optimize.cpp:76-137 ----------------------------------------------------
def optimize_continuous_v2(data, indicator, updating_factor, c_factor, updating_confd, gram, lam, tuning):
    u = np.array(updating_factor, float).copy()
    z = np.asarray(updating_confd, float)
    K = c_factor.shape[0]
    if tuning == 1:
        resid = data - np.outer(z, u @ c_factor)                                 # :84
        squared_factor = c_factor ** 2                                           # :88
        squared_confd = z ** 2
        norm_factor = squared_factor.sum(axis=1)                                 # :90
        zero_idx = [np.flatnonzero(indicator[k, :] == 0) for k in range(indicator.shape[0])]   # :92-95
        while True:
            pre = u.copy()                                                       # :103
            for i in range(K):
                resid = resid + u[i] * np.outer(z, c_factor[i, :])               # :107
                Xty = float(z @ (indicator * resid) @ c_factor[i, :])            # :111
                XtX = sum(squared_confd[k] * (norm_factor[i] - squared_factor[i, zero_idx[k]].sum())
                          for k in range(indicator.shape[0]))                    # :112-114
                u[i] = Xty / (XtX + lam)                                         # :117
                resid = resid - u[i] * np.outer(z, c_factor[i, :])               # :118
            if np.sum(np.abs(pre - u)) < 1e-1:                                   # :122
                break
    elif tuning == 0:
        Xty = c_factor @ data.T @ z                                              # :128
        XtX = (z @ z) * gram + lam * np.eye(K)                                   # :129-130
        u = np.linalg.solve(XtX, Xty)                                            # :131
    else:
        raise ValueError("Parameter tuning should be either 0 or 1!")
    return u


# -- src/optimize.cpp:200-253 ---------------------------------------------------
def optimize_col(data, indicator, row_factor, c_factor, lam, alpha, tuning, tol, seed=0, it=0, order_mode=0,
                 max_sweeps=10000, gene_offset=0):
    out = c_factor.copy()
    K = c_factor.shape[0]
    n, p = data.shape
    total_sweeps = 0
    if tuning == 1:
        gram = row_factor.T @ row_factor                             # :205
        cube = np.einsum("ia,ib->iab", row_factor, row_factor)       # :207-210
        for j in range(p):
            sel = np.flatnonzero(indicator[:, j])                    # :216
            feature = row_factor[sel, :]                             # :217
            XtX = gram - cube[indicator[:, j] == 0].sum(axis=0)      # :218-219
            outcome = data[sel, j]                                   # :220-221
            Xty = feature.T @ outcome                                # :222
            if alpha == 0.0:
                out[:, j] = solve_sympd(XtX + lam * np.eye(K), Xty)  # :224-226
            else:
                out[:, j], sw = strong_coordinate_descent(feature, outcome, c_factor[:, j], lam, alpha, XtX, Xty, tol,
                                                          seed, gene_offset + j, it, order_mode, max_sweeps)   # :228
                total_sweeps += sw
    elif tuning == 0:
        XtX = row_factor.T @ row_factor                              # :234
        Xty = row_factor.T @ data                                    # :235
        if alpha == 0.0:
            out = solve_sympd(XtX + lam * np.eye(K), Xty)            # :237-240
        else:
            for j in range(p):                                       # :245-247
                out[:, j], sw = strong_coordinate_descent(row_factor, data[:, j], c_factor[:, j], lam, alpha, XtX,
                                                          Xty[:, j], tol, seed, gene_offset + j, it, order_mode,
                                                          max_sweeps)
                total_sweeps += sw
    else:
        raise ValueError("Parameter tuning should be either 0 or 1!")
    return out, total_sweeps


# -- src/utils.cpp:56-102 -------------------------------------------------------
def evaluate(residual, train_mask, test_mask, tuning):
    if tuning == 0:
        s = float(np.sum(residual ** 2))
        return s, np.sqrt(s / residual.size), np.nan                 # :62-63 (test_rmse uninitialised)
    s = float(np.sum(residual[train_mask != 0] ** 2))                # :65
    tr = np.sqrt(s / np.count_nonzero(train_mask))                   # :66
    nte = np.count_nonzero(test_mask)
    te = np.sqrt(np.mean(residual[test_mask != 0] ** 2)) if nte else np.nan   # :67
    return s, tr, te


def compute_loss(cfd, column_factor, lam1, lam2, alpha, sum_residual):
    row_reg = sum(lam1 * np.linalg.norm(a, "fro") ** 2 for a in cfd)  # :83-86
    col_reg = lam2 * (1 - alpha) * np.linalg.norm(column_factor, "fro") ** 2   # :88
    l1_reg = lam2 * alpha * np.sum(np.abs(column_factor))            # :91
    return sum_residual / 2 + row_reg / 2 + col_reg / 2 + l1_reg, (sum_residual / 2, row_reg / 2, col_reg / 2, l1_reg)


# -- src/optimize.cpp:255-422 ---------------------------------------------------
def optimize(data, cfd_factors, column_factor, cfd_indicators, train_indicator, test_indicator, lam1=1.0, lam2=1.0,
             alpha=0.1, tuning=1, global_tol=1e-10, sub_tol=1e-5, max_iter=10000, seed=0, order_mode=0,
             max_sweeps=10000, ctns_confounder=None):
    data = np.asarray(data, float)
    n, p = data.shape
    cfd = [np.array(a, float).copy() for a in cfd_factors]
    Cf = np.array(column_factor, float).copy()
    ind = np.asarray(cfd_indicators).reshape(n, -1)
    c = ind.shape[1]
    Z = []                                                           # :294-313 one-hot index matrices
    for i in range(c):
        levels = np.unique(ind[:, i])
        z = np.zeros((n, len(levels)))
        for k, lv in enumerate(levels):
            z[ind[:, i] == lv, k] = 1.0
        Z.append(z)
    ctns = None if ctns_confounder is None else np.asarray(ctns_confounder, float).reshape(n, -1)
    def rows():
        rf = sum(cfd[i][ind[:, i] - 1, :] for i in range(c))
        return rf if ctns is None else rf + ctns @ cfd[c]                # :289,372
    row_factor = rows()                                              # :281-291
    residual = data - row_factor @ Cf                                # :320-321
    s, tr, te = evaluate(residual, train_indicator, test_indicator, tuning)   # :322
    loss, comps = compute_loss(cfd, Cf, lam1, lam2, alpha, s)        # :323
    traj = [(-1, tr, te, *comps, loss, np.nan, 1.0)]
    decay = 1.0
    it = 0
    total_sweeps = 0
    while it <= max_iter:                                            # :325
        gram = Cf @ Cf.T                                             # :332
        for i in range(c):                                           # :335
            residual = residual + Z[i] @ cfd[i] @ Cf                 # :338
            cfd[i] = optimize_row(residual, train_indicator, cfd[i], Cf, ind[:, i], gram, lam1, tuning)   # :339
            if i != c - 1 or ctns is not None:
                residual = residual - Z[i] @ cfd[i] @ Cf             # :353-355
        if ctns is not None:                                         # :340-351
            for j in range(ctns.shape[1]):
                residual = residual + np.outer(ctns[:, j], cfd[c][j, :] @ Cf)        # :344
                cfd[c][j, :] = optimize_continuous_v2(residual, train_indicator, cfd[c][j, :], Cf, ctns[:, j], gram,
                                                      lam1, tuning)                  # :345-346
                if j != ctns.shape[1] - 1:
                    residual = residual - np.outer(ctns[:, j], cfd[c][j, :] @ Cf)    # :347-349
        row_factor = rows()                                          # :365-373
        Cf, sw = optimize_col(data, train_indicator, row_factor, Cf, lam2, alpha, tuning, sub_tol * decay, seed, it,
                              order_mode, max_sweeps)                # :376
        total_sweeps += sw
        residual = data - row_factor @ Cf                            # :377-378
        if it % 10 == 0:                                             # :381
            pre_loss = loss
            s, tr, te = evaluate(residual, train_indicator, test_indicator, tuning)
            loss, comps = compute_loss(cfd, Cf, lam1, lam2, alpha, s)
            delta = pre_loss - loss
            for thr in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1):         # :389-403
                if delta / 1000 <= thr:
                    decay = thr
                    break
            else:
                decay = 1.0
            traj.append((it, tr, te, *comps, loss, delta, decay))
            if (pre_loss - loss) / pre_loss < global_tol:            # :405
                break
        it += 1
    return dict(row_matrices=cfd, column_factor=Cf, train_rmse=tr, test_rmse=te, loss=loss,
                traj=np.array(traj, float), iters=it, total_sweeps=total_sweeps)
