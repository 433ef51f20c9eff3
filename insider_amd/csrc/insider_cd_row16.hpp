// insider_cd_row16.hpp — "row16" elastic-net coordinate descent, four genes per wavefront, Gram matrix in LDS: the cross-check
// of the register-resident kernel (option cd_variant = 2) for K <= 32 (SLOTS = 1, 2) and for 32 < K <= 48 (SLOTS = 3: two waves'
// blocks still fit a CU's LDS), the solver for 32 < K <= 48 without an l1 term, and the evaluation pass (loss statistics) of the
// three-slot register kernel.  (Mid-round 4 it was THE sweep kernel for 32 < K <= 48: 16 outer-iterations/s at K = 40, c3's shape;
// the register kernel's three-slot form, insider_cd_reg.hpp, runs 76.  Also measured here and dropped: the register kernel's
// scaled state with the LDS reads issued a step ahead: +4 %; this kernel is bound by its occupancy, two waves per CU.)
//
// The sweep loop of strong_coordinate_descent (src/coordinate_descent.cpp:86-114) is a K-step sequential
// recurrence per gene and, at BASELINE's tolerances, runs for hundreds to thousands of sweeps: it is issue-bound
// (~4.6 SIMD cycles per vector instruction on gfx950; an LDS-crossbar ds_bpermute ~24, a v_readlane ~12).  This
// variant packs FOUR genes into a wave — gene g owns the 16-lane DPP row g, lane i of the row owns sweep positions
// i (slot 0) and 16 + i (slot 1) — and broadcasts the coordinate increment with DPP row_newbcast (2 moves, no LDS,
// no SGPR round trip).  DPP needs the source lane at compile time, so the per-lane STATE (h, beta) is physically
// permuted at the start of every sweep: position t of the sweep's order then sits in lane t % 16, slot t / 16, and
// the K steps are an unrolled straight-line sequence.  The Gram matrix is never permuted: it stays in LDS (zero
// diagonal, row pitch K) and every lane addresses the columns of the coordinates it currently holds; the row of
// step t comes from the sweep's order held in SGPRs (scalar loads from the order table).
// Per step and wave (4 genes): 13 vector + 2 LDS + ~4 scalar instructions for K > 16, 11 + 1 for K <= 16.
#pragma once

namespace insider {

template <int N>
__device__ __forceinline__ double row_bcast(double v)   // lane N of every 16-lane row to the whole row
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row16_max(double v)
{
    v = fmax(v, dpp_mov_d<0xB1>(v));
    v = fmax(v, dpp_mov_d<0x4E>(v));
    v = fmax(v, dpp_mov_d<0x141>(v));
    v = fmax(v, dpp_mov_d<0x140>(v));
    return v;
}

// LDS doubles per wave: 4 Gram blocks of K*K (zero diagonal, pitch K) + per gene NC diagonal entries, NC effective
// 1/(XtX_kk + l2) (0 = screened out), NC staging slots; NC = 32 up to K = 32, 64 beyond (round 4: SLOTS = 3, 4 for K <= 63)
__host__ __device__ inline int r16_nc(int K) { return K <= 32 ? 32 : 64; }
// doubles between the Gram blocks of the wave's four genes: K * K rounded up to 8 mod 32, i.e. the blocks are 64 bytes mod 256
// apart in LDS.  All genes follow the same coordinate order, so lane i of every row reads the SAME (row, column) of its
// gene's block in every step: with blocks a multiple of 256 bytes apart (K = 40: 12800) the four reads hit the same banks —
// SQ_LDS_BANK_CONFLICT was 69 % of the LDS cycles of the K = 40 kernel (round 4)
__host__ __device__ inline int r16_gstride(int K) { return ((K * K + 23) / 32) * 32 + 8; }
__host__ __device__ inline int r16_lds_doubles(int K) { return 4 * r16_gstride(K) + 3 * 4 * r16_nc(K); }

template <int SLOTS>
struct R16State {
    double h[SLOTS], beta[SLOTS], inv[SLOTS];
    int col[SLOTS];   // LDS byte offset of (gene's Gram block row 0, column of the coordinate held in this slot)
};

// One coordinate update at the static position T (src/coordinate_descent.cpp:91-110 in covariance form).
template <int SLOTS, int T>
__device__ __forceinline__ void r16_step(R16State<SLOTS> &S, const char *L, const uint32_t (&ordw)[8 * SLOTS], double la,
                                         int lane16)
{
    constexpr int s = T >> 4, it = T & 15;
    if constexpr (s < SLOTS) {
        // unconditional: positions beyond K and screened-out coordinates carry inv = beta = 0, i.e. a zero increment
        const int rowoff = (int)((ordw[T >> 1] >> (16 * (T & 1))) & 0xffffu);   // scalar: coordinate * pitch bytes
        double g[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) g[u] = *reinterpret_cast<const double *>(L + (S.col[u] + rowoff));
        const double cand = copysign(fmax(fabs(S.h[s]) - la, 0.0) * S.inv[s], S.h[s]);       // :94-104
        const double d = row_bcast<it>(cand - S.beta[s]);
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) S.h[u] = fma(-d, g[u], S.h[u]);                      // :106-107
        S.beta[s] = lane16 == it ? cand : S.beta[s];                                         // :108
    }
}

// acc[u] -= sum_{m < K} G[m][column held in slot u] * v_m, with v in COORDINATE order (slot m / 16 of lane m % 16)
template <int SLOTS, int M>
__device__ __forceinline__ void r16_gemv_step(double (&acc)[SLOTS], const double (&v)[SLOTS], const int (&col)[SLOTS],
                                              const char *L, int K, int pitchB)
{
    constexpr int s = M >> 4, it = M & 15;
    if constexpr (s < SLOTS) {
        if (M < K) {
            const double vm = row_bcast<it>(v[s]);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u)
                acc[u] = fma(-vm, *reinterpret_cast<const double *>(L + (col[u] + M * pitchB)), acc[u]);
        }
    }
}

#define R16_UNROLL32(F)                                                                                     \
    F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15) F(16) F(17) F(18)    \
    F(19) F(20) F(21) F(22) F(23) F(24) F(25) F(26) F(27) F(28) F(29) F(30) F(31)
// positions 32 .. 63 (slots 2 and 3; the steps are constexpr-guarded by `s < SLOTS`, so the two-slot instantiations lose nothing)
#define R16_UNROLL64(F)                                                                                     \
    R16_UNROLL32(F) F(32) F(33) F(34) F(35) F(36) F(37) F(38) F(39) F(40) F(41) F(42) F(43) F(44) F(45) F(46) \
    F(47) F(48) F(49) F(50) F(51) F(52) F(53) F(54) F(55) F(56) F(57) F(58) F(59) F(60) F(61) F(62) F(63)

template <int SLOTS>
__device__ __forceinline__ void r16_gemv(double (&acc)[SLOTS], const double (&v)[SLOTS], const int (&col)[SLOTS],
                                         const char *L, int K, int pitchB)
{
#define R16_G(M) r16_gemv_step<SLOTS, M>(acc, v, col, L, K, pitchB);
    R16_UNROLL64(R16_G)
#undef R16_G
}

// acc[u] += sum_{m < K} Mat[c_u * pitch + m] * v_m for the coordinate c_u = 16u + i of slot u (global matrix)
template <int SLOTS, int M>
__device__ __forceinline__ void r16_dense_mv_step(double (&acc)[SLOTS], const double (&v)[SLOTS],
                                                  const double *__restrict__ Mat, int pitch, int K, int i, bool ok)
{
    constexpr int s = M >> 4, it = M & 15;
    if constexpr (s < SLOTS) {
        if (M < K) {
            const double vm = row_bcast<it>(v[s]);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const int c = 16 * u + i;
                acc[u] += (ok && c < K ? Mat[c * pitch + M] : 0.0) * vm;
            }
        }
    }
}

// The solver.  lds: this wave's LDS block (r16_lds_doubles(K) doubles) with the four Gram blocks already loaded
// (zero diagonal).  q, Gll, beta: COORDINATE order (slot u of lane i of row g = coordinate 16u + i of gene g);
// beta = warm start in, solution out.  gene_ok: the row holds a gene.  Returns the row's sweep count (negated when the
// sweep cap, not convergence, ended the solve).
template <int SLOTS>
__device__ __forceinline__ int cd_row16(double *lds, int K, const double (&q)[SLOTS], const double (&Gll)[SLOTS],
                                        double (&beta)[SLOTS], bool gene_ok, const CdParams &P, int lane)
{
    const int row = lane >> 4, i = lane & 15;
    const int pitchB = K * 8;
    const char *L = reinterpret_cast<const char *>(lds);
    const int gbase = row * r16_gstride(K) * 8;                     // byte offset of this gene's Gram block
    const int NC = r16_nc(K);
    double *GllT = lds + 4 * r16_gstride(K) + row * NC, *invT = GllT + 4 * NC, *stage = invT + 4 * NC;
    const double la = P.lambda * P.alpha, l2 = P.lambda * (1.0 - P.alpha);
    const uint64_t rowmask = 0xffffull << (16 * row);
    bool valid[SLOTS];
    int cid[SLOTS];          // coordinate currently held by slot u
    double gl[SLOTS];        // its XtX_kk
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        valid[u] = gene_ok && 16 * u + i < K;
        cid[u] = 16 * u + i < K ? 16 * u + i : 0;
        gl[u] = valid[u] ? Gll[u] : 1.0;
    }
    // ---- strong rule and start values (:74-80), coordinate order ---------------------------------------------
    double aq = 0.0;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) aq = fmax(aq, valid[u] ? fabs(q[u]) : 0.0);
    const double thr = P.alpha * (2.0 * P.lambda - row16_max(aq));                        // :74
    R16State<SLOTS> S;
    bool active[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        active[u] = valid[u] && !(fabs(q[u]) < thr);
        S.beta[u] = active[u] ? beta[u] : 0.0;                                            // :78
        S.inv[u] = active[u] ? cd_rcp(gl[u] + l2) : 0.0;
        S.h[u] = valid[u] ? q[u] : 0.0;
        S.col[u] = gbase + cid[u] * 8;
        GllT[16 * u + i] = gl[u];
        invT[16 * u + i] = S.inv[u];
    }
    r16_gemv<SLOTS>(S.h, S.beta, S.col, L, K, pitchB);                                    // :79 h = q - offdiag(XtX) beta
    wave_sync();

    bool run = gene_ok;
    int sweep = 0, my_sweeps = 0;
    double bfin[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) bfin[u] = S.beta[u];
    int fcid[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) fcid[u] = cid[u];
    // coordinate ids of the next sweep's positions, fetched one sweep ahead
    int nc[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) nc[u] = 16 * u + i < K ? (int)P.order[16 * u + i] : 0;

    while (__any(run)) {
        // ---- this sweep's order: scalar copy for the row offsets, per-lane ids (prefetched) for the columns -----
        const uint32_t *orow = reinterpret_cast<const uint32_t *>(P.order + (size_t)(sweep & (int)(INSIDER_PERM_PERIOD - 1)) * ORDER_ROW + 64);
        uint32_t ordw[8 * SLOTS];
#pragma unroll
        for (int w = 0; w < 8 * SLOTS; ++w) ordw[w] = __builtin_amdgcn_readfirstlane((int)orow[w]);
        int nn[SLOTS];
        {
            const size_t nx = (size_t)((sweep + 1 < P.max_sweeps ? sweep + 1 : sweep) & (int)(INSIDER_PERM_PERIOD - 1)) * ORDER_ROW;
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) nn[u] = 16 * u + i < K ? (int)P.order[nx + 16 * u + i] : 0;
        }
        // ---- move the state to this sweep's positions (through LDS, indexed by coordinate) ---------------------
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) if (16 * u + i < K) stage[cid[u]] = S.h[u];
        wave_sync();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) S.h[u] = stage[nc[u]];
        wave_sync();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) if (16 * u + i < K) stage[cid[u]] = S.beta[u];
        wave_sync();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            cid[u] = nc[u];
            S.beta[u] = stage[cid[u]];
            gl[u] = GllT[cid[u]];
            S.inv[u] = 16 * u + i < K ? invT[cid[u]] : 0.0;
            S.beta[u] = 16 * u + i < K ? S.beta[u] : 0.0;
            S.col[u] = gbase + cid[u] * 8;
        }
        wave_sync();
        double beta0[SLOTS], g0[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) { beta0[u] = S.beta[u]; g0[u] = S.h[u] - S.beta[u] * gl[u]; }
        // ---- the sweep (:91-110) -----------------------------------------------------------------------------------
        // straight-line blocks of four steps (one scalar branch per block) so that the scheduler can hoist the
        // next steps' address arithmetic and LDS reads into the stalls of the dependent chain
#define R16_S(T) r16_step<SLOTS, T>(S, L, ordw, la, i);
#define R16_B(B) if (4 * (B) < K) { R16_S(4 * (B)) R16_S(4 * (B) + 1) R16_S(4 * (B) + 2) R16_S(4 * (B) + 3) }
        R16_B(0) R16_B(1) R16_B(2) R16_B(3) R16_B(4) R16_B(5) R16_B(6) R16_B(7)
        if constexpr (SLOTS > 2) { R16_B(8) R16_B(9) R16_B(10) R16_B(11) }
        if constexpr (SLOTS > 3) { R16_B(12) R16_B(13) R16_B(14) R16_B(15) }
#undef R16_B
#undef R16_S
        ++sweep;
        // ---- loss change of the sweep (:112-114), per gene ------------------------------------------------------------
        double term = 0.0, g1[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            g1[u] = S.h[u] - S.beta[u] * gl[u];
            const double db = S.beta[u] - beta0[u];
            const double t = -0.5 * db * (g0[u] + g1[u]) + 0.5 * l2 * (S.beta[u] * S.beta[u] - beta0[u] * beta0[u]) +
                             la * (fabs(S.beta[u]) - fabs(beta0[u]));
            term += 16 * u + i < K ? t : 0.0;
        }
        const double dloss = row16_sum(term);
        if (run) {
            bool finish = sweep >= P.max_sweeps;
            if (!finish && !(fabs(dloss) > P.tol)) {                                      // :114
                bool anyv = false;
#pragma unroll
                for (int u = 0; u < SLOTS; ++u) {                                         // :118-119 (grad = -g, beta = 0)
                    const bool viol = gene_ok && 16 * u + i < K && S.inv[u] == 0.0 && fabs(g1[u]) > P.alpha * P.lambda;
                    if (viol) { S.inv[u] = cd_rcp(gl[u] + l2); invT[cid[u]] = S.inv[u]; }  // :123
                    anyv = anyv || viol;
                }
                if ((__ballot(anyv) & rowmask) == 0) finish = true;                       // :120-121
            }
            if (finish) {   // park the row: zero increments from now on
                my_sweeps = (sweep >= P.max_sweeps && fabs(dloss) > P.tol) ? -sweep : sweep;   // negative: stopped by the cap
#pragma unroll
                for (int u = 0; u < SLOTS; ++u) {
                    bfin[u] = S.beta[u];
                    fcid[u] = cid[u];
                    S.beta[u] = 0.0;
                    S.inv[u] = 0.0;
                    if (16 * u + i < K) invT[cid[u]] = 0.0;
                }
                run = false;
            }
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) nc[u] = nn[u];
    }
    // ---- solution back to coordinate order ---------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) if (gene_ok && 16 * u + i < K) stage[fcid[u]] = bfin[u];
    wave_sync();
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) beta[u] = valid[u] ? stage[16 * u + i] : 0.0;
    wave_sync();
    return my_sweeps;
}

// ---------------------------------------------------------------------------------------------
// Kernel: column update with the row16 solver (K <= 48); same contract as k_cd_cols
// ---------------------------------------------------------------------------------------------
template <int SLOTS>
__global__ void __launch_bounds__(64) k_cd_cols_r16(ColArgs a)
{
    extern __shared__ double r16_lds[];
    const int lane = threadIdx.x;
    const int row = lane >> 4, i = lane & 15;
    const int K = a.K, KP = a.KP;
    const int slot = blockIdx.x * 4 + row;
    const int j = slot < a.p ? (a.gene_perm ? a.gene_perm[slot] : slot) : a.p;
    const bool gene = j < a.p;
    double *Gg = r16_lds + row * r16_gstride(K);
    const double *st = (a.stat && gene) ? a.stat + (size_t)j * a.stat_len : nullptr;
    // XtX_j = R'R - complement (src/optimize.cpp:218-219), or the shared R'R (:234); zero diagonal in LDS
    double q[SLOTS], Gll[SLOTS], beta[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = 16 * u + i;
        const bool ok = gene && c < K;
        Gll[u] = 1.0;
        for (int k = 0; k < K; ++k) {
            double v = 0.0;
            if (ok) {
                v = st ? st[stat_index(k, c)] : a.RtR[k * KP + c];
            }
            if (k == c) { Gll[u] = ok ? v : 1.0; v = 0.0; }
            if (c < K) Gg[k * K + c] = v;
        }
        q[u] = 0.0;
        beta[u] = 0.0;
        if (ok) {
            q[u] = a.Qfull[(size_t)j * KP + c];                                          // :222,235 via level sums
            if (st) q[u] -= st[stat_index(KP - 1, c)];                                   // minus the held-out part
            beta[u] = a.C[(size_t)j * KP + c];
        }
    }
    wave_sync();
    if (a.mode == COL_CD) {                                                              // :228,246
        int sweeps = cd_row16<SLOTS>(r16_lds, K, q, Gll, beta, gene, a.cd, lane);
        const bool capped = sweeps < 0;
        sweeps = capped ? -sweeps : sweeps;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u)
            if (gene && 16 * u + i < K) a.C[(size_t)j * KP + 16 * u + i] = beta[u];
        if (gene && i == 0) {
            if (a.cap_hits) {
                if (capped) atomicAdd(a.cap_hits, 1);
                if (sweeps > a.cap_hits[1]) atomicMax(a.cap_hits + 1, sweeps);
            }
            a.sweeps[j] = sweeps;
            if (a.sweep_bins) atomicAdd(&a.sweep_bins[blockIdx.x & 255], (unsigned long long)sweeps);
        }
    }
    if (!a.checkpoint) return;
    // ---- loss statistics with the (updated) column, coordinate order: fresh g = q - XtX beta -------------------------
    const char *L = reinterpret_cast<const char *>(r16_lds);
    int col[SLOTS];
    double g[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = 16 * u + i;
        col[u] = row * r16_gstride(K) * 8 + (c < K ? c : 0) * 8;
        g[u] = (gene && c < K) ? q[u] - Gll[u] * beta[u] : 0.0;
    }
    r16_gemv<SLOTS>(g, beta, col, L, K, K * 8);
    double t_bqg = 0.0, t_b2 = 0.0, t_b1 = 0.0, t_te = 0.0;
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = 16 * u + i;
        if (gene && c < K) {
            t_bqg += beta[u] * (q[u] + g[u]);
            t_b2 += beta[u] * beta[u];
            t_b1 += fabs(beta[u]);
        }
    }
    if (a.test_from_stats && st) {
        // sum_test (x - r'b)^2 = sum_held x^2 - 2 b'qc + b'(R'R b) - b'(q - g)      (see k_cd_cols)
        double rb[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) rb[u] = 0.0;
#define R16_D(M) r16_dense_mv_step<SLOTS, M>(rb, beta, a.RtR, KP, K, i, gene);
        R16_UNROLL64(R16_D)
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

// stand-alone batch form (insider_hip_strong_cd) on dense (XtX, Xty)
template <int SLOTS>
__global__ void __launch_bounds__(64)
k_cd_batch_r16(const double *__restrict__ XtX, const double *__restrict__ Xty, const double *__restrict__ wstart, int K,
               int64_t nprob, CdParams cd, double *__restrict__ beta_out, int *__restrict__ sweeps_out)
{
    extern __shared__ double r16_lds[];
    const int lane = threadIdx.x;
    const int row = lane >> 4, i = lane & 15;
    const int64_t b = (int64_t)blockIdx.x * 4 + row;
    const bool prob = b < nprob;
    double *Gg = r16_lds + row * r16_gstride(K);
    double q[SLOTS], Gll[SLOTS], beta[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int c = 16 * u + i;
        const bool ok = prob && c < K;
        Gll[u] = 1.0;
        for (int k = 0; k < K; ++k) {
            double v = ok ? XtX[(size_t)b * K * K + (size_t)k * K + c] : 0.0;
            if (k == c) { Gll[u] = ok ? v : 1.0; v = 0.0; }
            if (c < K) Gg[k * K + c] = v;
        }
        q[u] = ok ? Xty[(size_t)b * K + c] : 0.0;
        beta[u] = ok ? wstart[(size_t)b * K + c] : 0.0;
    }
    wave_sync();
    const int sw = cd_row16<SLOTS>(r16_lds, K, q, Gll, beta, prob, cd, lane);
#pragma unroll
    for (int u = 0; u < SLOTS; ++u)
        if (prob && 16 * u + i < K) beta_out[(size_t)b * K + 16 * u + i] = beta[u];
    if (prob && i == 0 && sweeps_out) sweeps_out[b] = sw < 0 ? -sw : sw;
}

}  // namespace insider
