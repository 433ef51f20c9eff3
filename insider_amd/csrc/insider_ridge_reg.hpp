// insider_ridge_reg.hpp — the alpha == 0 column update (src/optimize.cpp:224-226,237-240: solve(XtX + lambda I, Xty,
// likely_sympd) per gene) for K <= 32, register-resident: four genes per wavefront, same layout as the elastic-net
// kernel of insider_cd_reg.hpp (gene g = 16-lane DPP row g; lane i holds rows i and 16 + i of the gene's K x K system
// and the matching right-hand-side entries).  Gauss-Jordan elimination without pivoting (on an SPD matrix the pivots
// are the d_j of L D L', all positive, and the elimination is as stable as Cholesky): at step j the pivot row's
// entries reach the other lanes of the row as the DPP row_newbcast source of a 64-bit v_fmac_f64, one instruction per
// updated element and slot, no LDS, no barriers, one reciprocal per column.  ~1000 vector instructions per four genes
// at K = 30, against a wave-level LDS Cholesky per gene (k_ridge_cols) before.
#pragma once

namespace insider {

// d += bcast_IT(s) * m  /  d += bcast_IT(d) * m  /  bcast_IT(s): lane IT of every 16-lane row.
// s_nop 1 in each: a DPP read needs two wait states after a VALU write of its source register; the compiler's hazard
// recognizer does not look into inline asm, and it may place a register copy of the source right in front of the
// statement (observed: v_mov_b64 then the DPP read of the copy, wrong results).
template <int IT>
__device__ __forceinline__ void fmac_bcast(double &d, double s, double m)
{
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(d) : "v"(s), "v"(m), "n"(IT));
}
template <int IT>
__device__ __forceinline__ void fmac_bcast_self(double &d, double m)
{
    asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(m), "n"(IT));
}
template <int IT>
__device__ __forceinline__ double mov_bcast(double s)
{
    double d;
    asm volatile("s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(s), "n"(IT));
    return d;
}

// One elimination step with the static pivot J.  G[u][c] = element (16u + i, c) of the row's system; b[u] its
// right-hand side.  Rows beyond K are zero and never pivot; columns beyond K are zero.
template <int SLOTS, int KMAX, int J>
__device__ __forceinline__ void ridge_step(double (&G)[SLOTS][KMAX], double (&b)[SLOTS], double (&diag)[SLOTS], bool &ok,
                                           int K, int i)
{
    if constexpr (J < KMAX) {
        if (J < K) {                                       // wave-uniform
            constexpr int s = J >> 4, it = J & 15;
            const double piv = mov_bcast<it>(G[s][J]);
            ok = ok && piv > 0.0;
            const double nrp = -1.0 / piv;
            double nf[SLOTS];                              // minus the multiplier of this lane's rows; 0 for the pivot row
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) nf[u] = (u == s && i == it) ? 0.0 : G[u][J] * nrp;
            diag[s] = i == it ? piv : diag[s];
#pragma unroll
            for (int c = J + 1; c < KMAX; ++c) {
#pragma unroll
                for (int u = 0; u < SLOTS; ++u)
                    if (u != s) fmac_bcast<it>(G[u][c], G[s][c], nf[u]);
                fmac_bcast_self<it>(G[s][c], nf[s]);       // the slot holding the pivot row last: it is the DPP source
            }
#pragma unroll
            for (int u = 0; u < SLOTS; ++u)
                if (u != s) fmac_bcast<it>(b[u], b[s], nf[u]);
            fmac_bcast_self<it>(b[s], nf[s]);
        }
    }
}

template <int SLOTS, int KMAX>
__device__ __forceinline__ bool ridge_solve_regs(double (&G)[SLOTS][KMAX], double (&b)[SLOTS], int K, int i)
{
    bool ok = true;
    double diag[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) diag[u] = 1.0;
#define RIDGE_S(J) ridge_step<SLOTS, KMAX, J>(G, b, diag, ok, K, i);
    R16_UNROLL32(RIDGE_S)
#undef RIDGE_S
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) b[u] /= diag[u];
    return ok;
}

// XtX_j = R'R - complement (src/optimize.cpp:218-219) or the shared R'R (:234), rows 16u + i, all columns
template <int SLOTS, int KMAX>
__device__ __forceinline__ void ridge_load(const RidgeArgs &a, const double *st, bool gene, int i, double (&G)[SLOTS][KMAX])
{
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = 16 * u + i;
        const bool okc = gene && c < a.K;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int bk = k >> 4;
            const int si = bk >= u ? (bk * (bk + 1) / 2 + u) * 256 + (k & 15) * 16 + i
                                   : (u * (u + 1) / 2 + bk) * 256 + i * 16 + (k & 15);
            const double v = st ? st[si] : a.RtR[k * a.KP + c];   // the record holds XtX_j itself
            G[u][k] = (okc && k < a.K) ? v : 0.0;
        }
    }
}

template <int SLOTS, int KMAX>
__global__ void __launch_bounds__(64, 2) k_ridge_cols_reg(RidgeArgs a)   // 256 VGPRs: the whole system stays in registers
{
    const int lane = threadIdx.x;
    const int row = lane >> 4, i = lane & 15;
    const int K = a.K, KP = a.KP;
    const int j = blockIdx.x * 4 + row;
    const bool gene = j < a.p;
    const double *st = (a.stat && gene) ? a.stat + (size_t)j * a.stat_len : nullptr;
    double G[SLOTS][KMAX], q[SLOTS], beta[SLOTS];
    ridge_load<SLOTS, KMAX>(a, st, gene, i, G);
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = 16 * u + i;
        q[u] = 0.0;
        beta[u] = 0.0;
        if (gene && c < K) {
            q[u] = a.Qfull[(size_t)j * KP + c];                                          // :222,235 via level sums
            if (st) q[u] -= st[stat_index(KP - 1, c)];                                   // minus the held-out part
            beta[u] = a.C[(size_t)j * KP + c];
        }
    }
    if (a.solve) {
        double b[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            b[u] = q[u];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) G[u][k] += (k == 16 * u + i && gene && k < K) ? a.lambda : 0.0;   // :224,237
        }
        const bool ok = ridge_solve_regs<SLOTS, KMAX>(G, b, K, i);                       // :226,240
        // not positive definite: the general route of solve(..., likely_sympd) runs in k_ridge_cols for the marked genes
        if (gene && !ok && i == 0) { a.mark[j] = 1; *a.retry = 1; }
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            if (gene && ok) beta[u] = b[u];
            if (gene && 16 * u + i < K) a.C[(size_t)j * KP + 16 * u + i] = beta[u];
        }
        if (a.checkpoint) ridge_load<SLOTS, KMAX>(a, st, gene, i, G);                    // the elimination destroyed XtX
    }
    if (!a.checkpoint) return;
    // ---- loss statistics with the (updated) column: g = q - XtX beta -------------------------------------------------
    double g[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) g[u] = q[u];
    reg_gemv<SLOTS, KMAX>(g, beta, G, K);
    double t_bqg = 0.0, t_b2 = 0.0, t_b1 = 0.0, t_te = 0.0;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        if (gene && 16 * u + i < K) {
            t_bqg += beta[u] * (q[u] + g[u]);
            t_b2 += beta[u] * beta[u];
            t_b1 += fabs(beta[u]);
        }
    }
    if (a.test_from_stats && st) {   // as in k_cd_cols
        double rb[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) rb[u] = 0.0;
#define R16_D(M) r16_dense_mv_step<SLOTS, M>(rb, beta, a.RtR, KP, K, i, gene);
        R16_UNROLL32(R16_D)
#undef R16_D
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int c = 16 * u + i;
            if (gene && c < K) t_te += beta[u] * (rb[u] - (q[u] - g[u]) - 2.0 * st[stat_index(KP - 1, c)]);
        }
    }
    const double bqg = row16_sum(t_bqg), sb2 = row16_sum(t_b2), sb1 = row16_sum(t_b1), te = row16_sum(t_te);
    if (gene && i == 0) {
        a.sse_train[j] = a.yy[j] - bqg;
        a.b2[j] = sb2;
        a.b1[j] = sb1;
        if (a.test_from_stats) a.sse_test[j] = st ? st[stat_index(KP - 1, KP - 1)] + te : 0.0;
    }
}

}  // namespace insider
