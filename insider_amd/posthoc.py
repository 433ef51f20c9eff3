"""Post-hoc per-interaction-level regression of the reference (R/glm_interaction.R:2-30): for every level of the
interaction indicator, the residual rows of its samples are regressed on the column factor,
``glm(response ~ . - 1, family = gaussian())`` with response = the stacked residual rows and features = t(column_factor)
repeated once per sample; coefficients and two-sided t-test p-values are returned per level.

Closed form of that stacked least-squares problem (m samples in the level, p genes, K latent dimensions):
    beta = (C C')^-1 C mean_k(residual[k, :]),   RSS = sum_k ||residual[k, :] - C' beta||^2,
    Var(beta) = RSS / (m p - K) * (m C C')^-1,   p-value = 2 * P(T_{m p - K} > |beta / se|).
Downstream analysis on K x K systems: plain numpy on the host (nothing here is on the factorisation's hot path).
"""
import numpy as np


def glm_interaction(residual, train_indicator, interaction_indicator, column_factor, tol=1e-10, n_cores=10):
    """-> (coeff_matrix, pval_matrix), each (#levels) x K, row i-1 for level i (R/glm_interaction.R:4-5,26-27).
    ``train_indicator``, ``tol`` and ``n_cores`` are accepted and unused, exactly like the reference's signature."""
    from scipy import stats
    residual = np.asarray(residual, dtype=np.float64)
    Cm = np.asarray(column_factor, dtype=np.float64)
    ind = np.asarray(interaction_indicator).ravel()
    K, p = Cm.shape
    levels = np.unique(ind)
    coeff = np.zeros((len(levels), K))
    pval = np.zeros((len(levels), K))
    G = Cm @ Cm.T
    Ginv = np.linalg.inv(G)
    for i in levels:
        ids = np.flatnonzero(ind == i)
        m = ids.size
        beta = Ginv @ (Cm @ residual[ids].mean(axis=0))
        rss = float(np.sum((residual[ids] - beta @ Cm) ** 2))
        dof = m * p - K
        se = np.sqrt(rss / dof * np.diag(Ginv) / m)
        coeff[int(i) - 1] = beta
        pval[int(i) - 1] = 2.0 * stats.t.sf(np.abs(beta / se), dof)
    return coeff, pval
