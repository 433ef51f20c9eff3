// insider_kernels.hpp — gfx950 (CDNA4, wave64) device code of the INSIDER factorisation core.
//
// Formulation.  The reference forms, for every gene j,
//     XtX_j = R'R - sum_{i held out of gene j} r_i r_i'        src/optimize.cpp:218-219
//     Xty_j = sum_{i in train} x_ij r_i                         src/optimize.cpp:220-222
// and symmetrically per sample for the row update (src/optimize.cpp:162-171).  Both are "complement"
// statistics over the HELD-OUT entries of one contiguous line of the data matrix: with the augmented
// vector f~_e = [f_e (K entries), x_e] of a held-out element e,
//     S = sum_e f~_e f~_e'   (a (K+1) x (K+1) SYRK on gathered rows)
// holds the Gram complement (top-left K x K), the XtY complement (row K) and sum x^2 (corner).
// One wavefront owns one line: it streams the line's fp64 values and uint8 codes with coalesced,
// non-temporal 16 B/lane loads, compacts the held-out elements into an LDS ring with ballot/popcount, and
// drains the ring four elements at a time through v_mfma_f64_16x16x4_f64 (the SYRK is the one dense
// contraction of the path; fp64 MFMA peak == fp64 vector peak on MI355X, but the matrix pipe needs two
// operand registers per 2048 flops where v_fma_f64 would need an LDS operand per 2 flops).
// The elastic-net coordinate sweep is a wavefront-level vector loop in covariance form on the K x K
// Gram staged in LDS (one wave per subproblem, lane l owns coordinate l; no MFMA).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/insider_perm.h"

namespace insider {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;
constexpr int LIST_CAP = 256;   // held-out ring entries per wave (>= 2 * 128)
constexpr int CHUNK = 128;      // elements per streaming step (2 per lane); line pitches are multiples of it

constexpr int CODE_TRAIN = 1;   // bit 0 of a mask code: entry is in the train set
constexpr int CODE_TEST = 2;    // bit 1: entry is in the test set (neither bit: NA)

template <int NB>
struct Geo {
    static constexpr int KP = 16 * NB;               // padded row length of factor rows (>= K + 1)
    static constexpr int NBLK = NB * (NB + 1) / 2;   // lower-triangular 16x16 blocks
    static constexpr int STAT = NBLK * 256;          // doubles per unit of block-stored statistics
};

// ---------------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
    // waves never share LDS regions here; this only orders one wave's own cross-lane LDS traffic
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double readlane_d(double v, int srclane /* wave-uniform */)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// ---------------------------------------------------------------------------------------------
// streaming + compaction of one line
// ---------------------------------------------------------------------------------------------
// Streams `pitch` (multiple of CHUNK) elements of one line, appends those whose code passes the filter to the
// wave's LDS ring (element index + value) in ascending element order and calls drain(ngroups) whenever the ring
// could overflow, and once at the end after padding to a multiple of GROUP with (-1, 0.0) dummies.
//   FILTER 0: held out of the train set (code & CODE_TRAIN) == 0      FILTER 1: test entries (code & CODE_TEST)
template <int FILTER, int GROUP, class Drain>
__device__ __forceinline__ void stream_line(const double *__restrict__ vals, const uint8_t *__restrict__ codes,
                                            int pitch, int *li, double *lx, int lane, Drain &&drain)
{
    int cnt = 0;
    const uint64_t lt = lanemask_lt(lane);
    for (int base = 0; base < pitch; base += CHUNK) {
        const int e0 = base + 2 * lane;
        d2 xv = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vals + e0));
        uint16_t cw = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(codes + e0));
        const int c0 = cw & 0xff, c1 = cw >> 8;
        const bool h0 = FILTER == 0 ? !(c0 & CODE_TRAIN) : (c0 & CODE_TEST) != 0;
        const bool h1 = FILTER == 0 ? !(c1 & CODE_TRAIN) : (c1 & CODE_TEST) != 0;
        const uint64_t b0 = __ballot(h0), b1 = __ballot(h1);
        if (b0 | b1) {
            const int n0 = __popcll(b0);
            if (h0) { const int pos = cnt + __popcll(b0 & lt); li[pos] = e0; lx[pos] = xv.x; }
            if (h1) { const int pos = cnt + n0 + __popcll(b1 & lt); li[pos] = e0 + 1; lx[pos] = xv.y; }
            cnt += n0 + __popcll(b1);
            if (cnt > LIST_CAP - CHUNK) {
                wave_sync();
                const int ng = cnt / GROUP;
                drain(ng);
                const int rem = cnt - ng * GROUP;   // < GROUP <= 64
                int ti = 0; double tx = 0.0;
                if (lane < rem) { ti = li[ng * GROUP + lane]; tx = lx[ng * GROUP + lane]; }
                wave_sync();
                if (lane < rem) { li[lane] = ti; lx[lane] = tx; }
                cnt = rem;
            }
        }
    }
    const int padded = (cnt + GROUP - 1) / GROUP * GROUP;
    for (int i = cnt + lane; i < padded; i += WAVE) { li[i] = -1; lx[i] = 0.0; }
    wave_sync();
    if (padded) drain(padded / GROUP);
    wave_sync();
}

// SYRK drain: four ring entries per step, lane l feeds row (l >> 4) of the 4-deep panel, column (l & 15) of
// each 16-wide block.  A and B operands of v_mfma_f64_16x16x4_f64 are the same registers (A[i][k] = f~_k[16bi+i],
// B[k][j] = f~_k[16bj+j]), so a step costs NB gathered loads and NB(NB+1)/2 MFMAs.
template <int NB>
__device__ __forceinline__ void drain_syrk(const int *li, const double *lx, int ngroups, const double *__restrict__ F,
                                           int K, d4 (&acc)[Geo<NB>::NBLK], int lane)
{
    constexpr int KP = Geo<NB>::KP;
    const int sub = lane >> 4, c16 = lane & 15;
#pragma unroll 2
    for (int g = 0; g < ngroups; ++g) {
        const int e = li[4 * g + sub];
        const double xv = lx[4 * g + sub];
        double a[NB];
        const double *row = F + (size_t)(e < 0 ? 0 : e) * KP;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int col = 16 * b + c16;
            double v = row[col];
            v = e < 0 ? 0.0 : v;
            a[b] = col == K ? xv : v;
        }
        int blk = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj, ++blk)
                acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[blk], 0, 0, 0);
    }
}

// v_mfma_f64_16x16x4_f64 result map: register r of lane l holds D[(l >> 4) + 4 r][l & 15].
// Scatter the lower blocks into a full symmetric KP x KP LDS matrix.
template <int NB>
__device__ __forceinline__ void acc_to_lds(const d4 (&acc)[Geo<NB>::NBLK], double *G, int lane)
{
    constexpr int KP = Geo<NB>::KP;
    const int sub = lane >> 4, c16 = lane & 15;
    int blk = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj, ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 16 * bi + sub + 4 * r, b = 16 * bj + c16;
                const double v = acc[blk][r];
                G[a * KP + b] = v;
                G[b * KP + a] = v;
            }
}

// ---------------------------------------------------------------------------------------------
// elastic-net coordinate descent, covariance form, one wave per subproblem
// (strong_coordinate_descent, src/coordinate_descent.cpp:56-127)
// ---------------------------------------------------------------------------------------------
struct CdParams {
    double lambda, alpha, tol;
    uint64_t seed;
    uint32_t iter;
    int max_sweeps;
    int order_mode;
};

// G: LDS, full symmetric, row pitch KP (>= K).  Lane l < K owns coordinate l: q = Xty_l, beta = warm start in,
// solution out.  g_out = Xty - XtX beta (all lanes < K).  Returns the number of sweeps.
//   reference residual form              covariance form used here
//   dot(residual, X_k)          :94  ==  g_k = (Xty - XtX beta)_k
//   residual -= d X_k          :107  ==  g -= d XtX[:,k]
//   |pre_loss - iter_loss|     :114  ==  |sum over the sweep's updates of the exact loss change|
__device__ __forceinline__ int cd_solve(const double *G, int KP, int K, double q, double &beta, double &g_out,
                                        const CdParams &P, uint32_t unit, int lane)
{
    const bool valid = lane < K;
    const double la = P.lambda * P.alpha, l2 = P.lambda * (1.0 - P.alpha);
    const double aq = valid ? fabs(q) : 0.0;
    const double mx = wave_max(aq);
    const double thr = P.alpha * (2.0 * P.lambda - mx);                                // :74
    bool active = valid && !(aq < thr);
    if (!active) beta = 0.0;                                                           // :78
    double g = valid ? q : 0.0;                                                        // :79 (as Xty - XtX beta)
    for (int m = 0; m < K; ++m) {
        const double bm = readlane_d(beta, m);
        if (bm != 0.0) g -= (valid ? G[m * KP + lane] : 0.0) * bm;
    }
    const double Gll = valid ? G[lane * KP + lane] : 1.0;
    const double inv = 1.0 / (Gll + l2);
    const uint64_t lt = lanemask_lt(lane);
    int sweep = 0;
    for (;;) {
        const uint64_t amask = __ballot(active);                                       // :83
        const int na = __popcll(amask);
        double dl;
        do {
            int rank;
            if (P.order_mode == 0) {                                                   // :89 randperm -> hashed order
                const uint32_t base = insider_perm_base(P.seed, unit, P.iter, (uint32_t)sweep);
                const uint32_t key = insider_perm_key(base, (uint32_t)lane);
                rank = 0;
                for (uint64_t m = amask; m; m &= m - 1) {
                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(m));
                    rank += kb < key;
                }
            } else {
                rank = __popcll(amask & lt);
            }
            ++sweep;
            dl = 0.0;
            for (int t = 0; t < na; ++t) {                                             // :91
                const int k = __builtin_ctzll(__ballot(active && rank == t));
                const double gcol = valid ? G[k * KP + lane] : 0.0;
                const double gk = readlane_d(g, k), bk = readlane_d(beta, k);
                const double Gkk = readlane_d(Gll, k), ik = readlane_d(inv, k);
                const double u = gk + bk * Gkk;                                        // :94
                const double au = fabs(u);
                const double nb = au > la ? copysign(au - la, u) * ik : 0.0;           // :99-104
                if (nb != bk) {                                                        // :106
                    const double d = nb - bk;
                    g -= d * gcol;                                                     // :107
                    if (lane == k) beta = nb;                                          // :108
                    dl += d * (0.5 * d * Gkk - gk) + 0.5 * l2 * (nb * nb - bk * bk) + la * (fabs(nb) - fabs(bk));
                }
            }
        } while (fabs(dl) > P.tol && sweep < P.max_sweeps);                            // :114
        if (sweep >= P.max_sweeps) break;
        const bool viol = valid && !active && fabs(g) > P.alpha * P.lambda;            // :118-119 (grad = -g)
        if (!__any(viol)) break;                                                       // :120
        active = active || viol;                                                       // :123
    }
    g_out = g;
    return sweep;
}

// ---------------------------------------------------------------------------------------------
// wave-level Cholesky on an LDS matrix: solve(XtX, Xty, likely_sympd)
// (src/optimize.cpp:175,190,226,240).  A: full symmetric, row pitch KP; destroyed (L in the lower triangle,
// L' mirrored in the upper).  b: lane l < K holds b_l in, x_l out.  Returns false if not positive definite.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool chol_solve_lds(double *A, int KP, int K, double &b, int lane)
{
    const bool valid = lane < K;
    for (int j = 0; j < K; ++j) {
        const double d = A[j * KP + j];
        if (!(d > 0.0)) return false;
        const double s = sqrt(d);
        const bool below = valid && lane > j;
        double lij = 0.0;
        if (below) lij = A[j * KP + lane] / s;
        wave_sync();
        if (lane == j) A[j * KP + j] = s;
        if (below) { A[j * KP + lane] = lij; A[lane * KP + j] = lij; }
        wave_sync();
        for (int m = j + 1; m < K; ++m) {
            const double lmj = A[j * KP + m];
            if (below) A[m * KP + lane] -= lmj * lij;
        }
        wave_sync();
    }
    for (int j = 0; j < K; ++j) {           // L y = b
        const double yj = readlane_d(b, j) / A[j * KP + j];
        if (lane == j) b = yj;
        if (valid && lane > j) b -= A[j * KP + lane] * yj;
    }
    for (int j = K - 1; j >= 0; --j) {      // L' x = y
        const double xj = readlane_d(b, j) / A[j * KP + j];
        if (lane == j) b = xj;
        if (lane < j) b -= A[j * KP + lane] * xj;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Kernel: complement statistics of every line (row side; also the column side's stand-alone form)
// ---------------------------------------------------------------------------------------------
// unit u, segment s: elements [s * seg_len, (s+1) * seg_len) of line u.  Output: Geo<NB>::STAT doubles at
// stat[(s * units + u) * STAT], block-major [blk][16][16] of the lower blocks of sum f~ f~'.
template <int NB, int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_line_stats(const double *__restrict__ vals, const uint8_t *__restrict__ codes, int64_t pitch, int units, int nseg,
             int seg_len, const double *__restrict__ F, int K, double *__restrict__ stat)
{
    constexpr int NBLK = Geo<NB>::NBLK;
    __shared__ int s_li[WPB][LIST_CAP];
    __shared__ double s_lx[WPB][LIST_CAP];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * WPB + w;
    if (item >= (int64_t)units * nseg) return;
    const int u = (int)(item % units), s = (int)(item / units);
    d4 acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    const int64_t off = (int64_t)u * pitch + (int64_t)s * seg_len;
    const int64_t left = pitch - (int64_t)s * seg_len;
    const int this_len = left < seg_len ? (int)left : seg_len;   // both multiples of CHUNK
    const double *Fs = F + (size_t)s * seg_len * Geo<NB>::KP;
    int *li = s_li[w];
    double *lx = s_lx[w];
    stream_line<0, 4>(vals + off, codes + off, this_len, li, lx, lane,
                      [&](int ng) { drain_syrk<NB>(li, lx, ng, Fs, K, acc, lane); });
    double *out = stat + (size_t)item * Geo<NB>::STAT;
    const int sub = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[b * 256 + (sub + 4 * r) * 16 + c16] = acc[b][r];
}

// ---------------------------------------------------------------------------------------------
// Kernel: fused column update (optimize_col, src/optimize.cpp:200-253) + checkpoint statistics
// ---------------------------------------------------------------------------------------------
enum ColMode { COL_EVAL = 0, COL_CD = 1, COL_RIDGE = 2 };

struct ColArgs {
    const double *vals;      // X, gene-major lines of pitch ldn
    const uint8_t *codes;
    int64_t pitch;
    int p;
    int K;
    int masked;              // tuning == 1: stream the line; 0: shared Gram, no streaming
    const double *R;         // n x KP row factor rows (zero beyond K)
    const double *RtR;       // KP x KP
    const double *Qfull;     // p x KP: R' x_j over ALL samples (from the per-level sums)
    double *C;               // p x KP: warm start in, solution out
    const double *yy;        // p: sum of x^2 over train entries (masked) or all entries (unmasked)
    int mode;                // ColMode
    int checkpoint;          // also produce per-gene loss statistics
    CdParams cd;
    int64_t gene_offset;
    double *sse_train, *sse_test, *b2, *b1;   // p each (checkpoint only)
    int *sweeps;             // p
    int *fail;               // set to 1 if a ridge system was not positive definite
};

template <int NB, int WPB>
__global__ void __launch_bounds__(WPB * 64) k_col_update(ColArgs a)
{
    constexpr int KP = Geo<NB>::KP, NBLK = Geo<NB>::NBLK;
    __shared__ double s_G[WPB][KP * KP];
    __shared__ int s_li[WPB][LIST_CAP];
    __shared__ double s_lx[WPB][LIST_CAP];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * WPB + w;
    if (j >= a.p) return;
    const int K = a.K;
    double *G = s_G[w];
    int *li = s_li[w];
    double *lx = s_lx[w];
    const bool valid = lane < K;
    const int64_t off = (int64_t)j * a.pitch;

    double q = valid ? a.Qfull[(size_t)j * KP + lane] : 0.0;
    if (a.masked) {
        d4 acc[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
        stream_line<0, 4>(a.vals + off, a.codes + off, (int)a.pitch, li, lx, lane,
                          [&](int ng) { drain_syrk<NB>(li, lx, ng, a.R, K, acc, lane); });
        acc_to_lds<NB>(acc, G, lane);
        wave_sync();
        if (valid) q -= G[K * KP + lane];                                               // :220-222 as complement
        wave_sync();
        for (int i = lane; i < KP * KP; i += WAVE) G[i] = a.RtR[i] - G[i];              // :218-219
    } else {
        for (int i = lane; i < KP * KP; i += WAVE) G[i] = a.RtR[i];                     // :234
    }
    wave_sync();

    double beta = valid ? a.C[(size_t)j * KP + lane] : 0.0;
    double g = 0.0;
    int sweeps = 0;
    if (a.mode == COL_CD) {                                                             // :228,246
        sweeps = cd_solve(G, KP, K, q, beta, g, a.cd, (uint32_t)(a.gene_offset + j), lane);
        if (valid) a.C[(size_t)j * KP + lane] = beta;
    } else if (a.mode == COL_RIDGE) {                                                   // :224-226,237-240
        if (valid) G[lane * KP + lane] += a.cd.lambda;
        wave_sync();
        double b = q;
        const bool ok = chol_solve_lds(G, KP, K, b, lane);   // destroys G (rebuilt below if needed)
        if (!ok && lane == 0) *a.fail = 1;
        if (ok) beta = b;
        if (valid) a.C[(size_t)j * KP + lane] = beta;
    }
    if (lane == 0) a.sweeps[j] = sweeps;
    if (!a.checkpoint) return;

    // ---- loss statistics with the updated column (src/utils.cpp:56-102 evaluated per gene) ----
    if (a.mode == COL_RIDGE) {   // G was destroyed by the factorisation: rebuild it
        wave_sync();
        if (a.masked) {
            d4 acc[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
            stream_line<0, 4>(a.vals + off, a.codes + off, (int)a.pitch, li, lx, lane,
                              [&](int ng) { drain_syrk<NB>(li, lx, ng, a.R, K, acc, lane); });
            acc_to_lds<NB>(acc, G, lane);
            wave_sync();
            for (int i = lane; i < KP * KP; i += WAVE) G[i] = a.RtR[i] - G[i];
        } else {
            for (int i = lane; i < KP * KP; i += WAVE) G[i] = a.RtR[i];
        }
        wave_sync();
    }
    // fresh g = q - G beta (the incrementally updated one carries the sweeps' round-off)
    g = valid ? q : 0.0;
    for (int m = 0; m < K; ++m) {
        const double bm = readlane_d(beta, m);
        if (bm != 0.0) g -= (valid ? G[m * KP + lane] : 0.0) * bm;
    }
    // sum_train (x - r'beta)^2 = yy - 2 beta'q + beta'G beta = yy - beta'(q + g)
    const double bqg = wave_sum(valid ? beta * (q + g) : 0.0);
    const double sb2 = wave_sum(valid ? beta * beta : 0.0);
    const double sb1 = wave_sum(valid ? fabs(beta) : 0.0);
    double te = 0.0;
    if (a.masked) {   // test entries: explicit residuals on the compacted list
        stream_line<1, 64>(a.vals + off, a.codes + off, (int)a.pitch, li, lx, lane, [&](int ng) {
            for (int gi = 0; gi < ng; ++gi) {
                const int e = li[64 * gi + lane];
                const double xv = lx[64 * gi + lane];
                const double *row = a.R + (size_t)(e < 0 ? 0 : e) * KP;
                double dot = 0.0;
                for (int k = 0; k < K; ++k) dot += row[k] * readlane_d(beta, k);
                const double r = xv - dot;
                te += e < 0 ? 0.0 : r * r;
            }
        });
        te = wave_sum(te);
    }
    if (lane == 0) {
        a.sse_train[j] = a.yy[j] - bqg;
        a.sse_test[j] = te;
        a.b2[j] = sb2;
        a.b1[j] = sb1;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel: batched stand-alone CD from (XtX, Xty) in global memory (insider_hip_strong_cd)
// ---------------------------------------------------------------------------------------------
template <int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_cd_batch(const double *__restrict__ XtX, const double *__restrict__ Xty, const double *__restrict__ wstart, int K,
           int64_t nprob, CdParams cd, uint32_t unit0, double *__restrict__ beta_out, int *__restrict__ sweeps_out)
{
    __shared__ double s_G[WPB][64 * 64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * WPB + w;
    if (b >= nprob) return;
    double *G = s_G[w];
    const double *src = XtX + (size_t)b * K * K;
    for (int i = lane; i < K * K; i += WAVE) G[i] = src[i];
    wave_sync();
    const bool valid = lane < K;
    const double q = valid ? Xty[(size_t)b * K + lane] : 0.0;
    double beta = valid ? wstart[(size_t)b * K + lane] : 0.0, g;
    const int sw = cd_solve(G, K, K, q, beta, g, cd, unit0 + (uint32_t)b, lane);
    if (valid) beta_out[(size_t)b * K + lane] = beta;
    if (lane == 0 && sweeps_out) sweeps_out[b] = sw;
}

// ---------------------------------------------------------------------------------------------
// small dense kernels around the two streaming passes
// ---------------------------------------------------------------------------------------------

// R[r][:] = sum_i Astack[off_i + level_i(r)][:]   (src/optimize.cpp:365-369), rows of pitch KP
__global__ void k_build_R(const int *__restrict__ lev /*c x n, 0-based*/, const int *__restrict__ lvl_off, int c, int n,
                          const double *__restrict__ Astack, int KP, double *__restrict__ R)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * KP) return;
    const int r = (int)(t / KP), k = (int)(t % KP);
    double s = 0.0;
    for (int i = 0; i < c; ++i) s += Astack[(size_t)(lvl_off[i] + lev[(size_t)i * n + r]) * KP + k];
    R[t] = s;
}

// Partial Gram of a tall matrix F (rows x KP): part[blk] = sum over the block's rows of f f'  (KP x KP)
template <int KP>
__global__ void __launch_bounds__(256) k_gram_partial(const double *__restrict__ F, int64_t rows, int rows_per_blk,
                                                      double *__restrict__ part)
{
    __shared__ double s_f[64][KP + 1];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = r0 + rows_per_blk < rows ? r0 + rows_per_blk : rows;
    constexpr int PER = (KP * KP + 255) / 256;
    double acc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[i] = 0.0;
    for (int64_t rb = r0; rb < r1; rb += 64) {
        const int nr = (int)(r1 - rb < 64 ? r1 - rb : 64);
        __syncthreads();
        for (int i = threadIdx.x; i < nr * KP; i += 256) s_f[i / KP][i % KP] = F[(size_t)(rb + i / KP) * KP + i % KP];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int o = threadIdx.x + 256 * i;
            if (o < KP * KP) {
                const int x = o / KP, y = o % KP;
                double s = acc[i];
                for (int r = 0; r < nr; ++r) s += s_f[r][x] * s_f[r][y];
                acc[i] = s;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int o = threadIdx.x + 256 * i;
        if (o < KP * KP) part[(size_t)blockIdx.x * KP * KP + o] = acc[i];
    }
}

// out[o] = sum_b part[b][o]  (fixed order: bitwise reproducible)
__global__ void k_sum_partials(const double *__restrict__ part, int nblk, int len, double *__restrict__ out)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= len) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * len + o];
    out[o] = s;
}

// Qfull[j][k] = sum_l Astack[l][k] * S[j][l]     (== R' x_j over all samples: x summed per level first)
__global__ void __launch_bounds__(256) k_qfull(const double *__restrict__ S /*p x SLP*/, int SL, int SLP,
                                               const double *__restrict__ Astack /*SL x KP*/, int KP, int p,
                                               double *__restrict__ Q)
{
    constexpr int LT = 64;
    __shared__ double s_A[LT * 64];
    __shared__ double s_S[16 * LT];
    const int gpb = 256 / KP;   // genes per block
    const int k = threadIdx.x % KP, gl = threadIdx.x / KP;
    const int j = blockIdx.x * gpb + gl;
    const bool act = gl < gpb && j < p;
    double acc = 0.0;
    for (int l0 = 0; l0 < SL; l0 += LT) {
        const int nl = SL - l0 < LT ? SL - l0 : LT;
        __syncthreads();
        for (int i = threadIdx.x; i < nl * KP; i += 256) s_A[i] = Astack[(size_t)l0 * KP + i];
        for (int i = threadIdx.x; i < gpb * nl; i += 256) {
            const int g = i / nl, l = i % nl, jj = blockIdx.x * gpb + g;
            s_S[g * LT + l] = jj < p ? S[(size_t)jj * SLP + l0 + l] : 0.0;
        }
        __syncthreads();
        if (act)
            for (int l = 0; l < nl; ++l) acc += s_A[l * KP + k] * s_S[gl * LT + l];
    }
    if (act) Q[(size_t)j * KP + k] = acc;
}

// part[gb][l][k] = sum over gene block gb of S[j][l] * C[j][k]   (per-level sums of X C'); block (gb, level tile)
__global__ void __launch_bounds__(256) k_sc_partial(const double *__restrict__ S, int SL, int SLP,
                                                    const double *__restrict__ C, int KP, int p, int genes_per_blk,
                                                    double *__restrict__ part)
{
    __shared__ double s_C[32 * 64];
    __shared__ double s_S[32 * 16];
    const int LT = 256 / KP;
    const int l_loc = threadIdx.x / KP, k = threadIdx.x % KP;
    const int l = blockIdx.y * LT + l_loc;
    const bool act = l_loc < LT && l < SL;
    const int j0 = blockIdx.x * genes_per_blk;
    const int j1 = j0 + genes_per_blk < p ? j0 + genes_per_blk : p;
    double acc = 0.0;
    for (int jb = j0; jb < j1; jb += 32) {
        const int ng = j1 - jb < 32 ? j1 - jb : 32;
        __syncthreads();
        for (int i = threadIdx.x; i < ng * KP; i += 256) s_C[i] = C[(size_t)jb * KP + i];
        for (int i = threadIdx.x; i < ng * LT; i += 256) {
            const int g = i / LT, ll = i % LT, lg = blockIdx.y * LT + ll;
            s_S[g * LT + ll] = lg < SL ? S[(size_t)(jb + g) * SLP + lg] : 0.0;
        }
        __syncthreads();
        if (act)
            for (int g = 0; g < ng; ++g) acc += s_S[g * LT + l_loc] * s_C[g * KP + k];
    }
    if (act) part[((size_t)blockIdx.x * SL + l) * KP + k] = acc;
}

// ---------------------------------------------------------------------------------------------
// row update (optimize_row, src/optimize.cpp:139-198) from the per-sample complement statistics
// ---------------------------------------------------------------------------------------------
// For covariate i, level l, with s_r = sum_{m != i} A_m[level_m(r)] (the Gauss-Seidel residual of :338,354 is
// x_r - s_r' C) and per-sample complements Hc_r = sum_{held out} c c', bc_r = sum_{held out} x c:
//   XtX_l = cnt_l CC' - sum_r Hc_r (+ lambda I)                                          :170,174
//   Xty_l = (S_i C')[l] - sum_r bc_r - CC' (sum_r s_r) + sum_r Hc_r s_r                  :171
// Stage 1 (one wave per chunk of <= 16 member samples): the r-sums, in block layout.
struct LevelArgs {
    const double *stat;      // [nseg][n][STAT] complement statistics (masked) — may be null when !masked
    int nseg, n, K, masked;
    const int *lev;          // c x n, 0-based
    const int *lvl_off;      // c + 1 prefix offsets into the stacked levels
    int c, cov;              // updating covariate
    const int *chunk_level;  // per chunk: level (0-based within the covariate)
    const int *chunk_begin;  // per chunk: range into members
    const int *chunk_end;
    const int *members;      // sample ids sorted by level (this covariate)
    int nchunks;
    const double *Astack;    // SL x KP
    double *part;            // [nchunks][STAT + 2*KP]: blocks of sum Hc, then v = sum(Hc s - bc), then ssum = sum s
};

template <int NB>
__global__ void __launch_bounds__(64) k_level_partial(LevelArgs a)
{
    constexpr int KP = Geo<NB>::KP, NBLK = Geo<NB>::NBLK, STAT = Geo<NB>::STAT;
    __shared__ double s_H[KP * KP];
    __shared__ double s_s[KP];
    const int ch = blockIdx.x, lane = threadIdx.x;
    if (ch >= a.nchunks) return;
    const int K = a.K;
    const bool valid = lane < K;
    const int sub = lane >> 4, c16 = lane & 15;
    d4 hsum[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) hsum[b] = d4{0.0, 0.0, 0.0, 0.0};
    double v = 0.0, ssum = 0.0;
    for (int mi = a.chunk_begin[ch]; mi < a.chunk_end[ch]; ++mi) {
        const int r = a.members[mi];
        double s = 0.0;
        if (valid)
            for (int m = 0; m < a.c; ++m)
                if (m != a.cov) s += a.Astack[(size_t)(a.lvl_off[m] + a.lev[(size_t)m * a.n + r]) * KP + lane];
        ssum += s;
        if (a.masked) {
            d4 h[NBLK];
#pragma unroll
            for (int b = 0; b < NBLK; ++b) h[b] = d4{0.0, 0.0, 0.0, 0.0};
            for (int sg = 0; sg < a.nseg; ++sg) {
                const double *src = a.stat + ((size_t)sg * a.n + r) * STAT;
#pragma unroll
                for (int b = 0; b < NBLK; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) h[b][q] += src[b * 256 + (sub + 4 * q) * 16 + c16];
            }
            wave_sync();
            acc_to_lds<NB>(h, s_H, lane);
            if (lane < KP) s_s[lane] = valid ? s : 0.0;
            wave_sync();
            if (valid) {
                double y = -s_H[K * KP + lane];            // - bc_r
                for (int b = 0; b < K; ++b) y += s_H[lane * KP + b] * s_s[b];
                v += y;
            }
            // the augmented row/column K (bc, sum x^2) must not enter the Gram sum
#pragma unroll
            for (int b = 0; b < NBLK; ++b) hsum[b] += h[b];
        }
    }
    double *out = a.part + (size_t)ch * (STAT + 2 * KP);
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[b * 256 + (sub + 4 * q) * 16 + c16] = hsum[b][q];
    if (lane < KP) { out[STAT + lane] = valid ? v : 0.0; out[STAT + KP + lane] = valid ? ssum : 0.0; }
}

// Stage 2 (one wave per level): sum the level's chunk partials in fixed order and form this rank's share of
// the normal equations: eq[l] = {XtX (KP x KP, full symmetric, no ridge term), Xty (KP)}.
struct LevelReduceArgs {
    const double *part;
    const int *lvl_chunk_ptr;   // per level: chunk range
    const int *lvl_count;       // per level: member count
    int L, K;
    const double *CCt;          // KP x KP (this rank's gene slab)
    const double *SC;           // [SL][KP]: (S C') rows; this covariate starts at sc_off
    int sc_off;
    double *eq;                 // [L][KP*KP + KP]
};

template <int NB>
__global__ void __launch_bounds__(64) k_level_reduce(LevelReduceArgs a)
{
    constexpr int KP = Geo<NB>::KP, NBLK = Geo<NB>::NBLK, STAT = Geo<NB>::STAT;
    __shared__ double s_H[KP * KP];
    __shared__ double s_s[KP];
    const int l = blockIdx.x, lane = threadIdx.x;
    if (l >= a.L) return;
    const int K = a.K;
    const bool valid = lane < K;
    const int sub = lane >> 4, c16 = lane & 15;
    d4 h[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) h[b] = d4{0.0, 0.0, 0.0, 0.0};
    double v = 0.0, ssum = 0.0;
    for (int ch = a.lvl_chunk_ptr[l]; ch < a.lvl_chunk_ptr[l + 1]; ++ch) {
        const double *src = a.part + (size_t)ch * (STAT + 2 * KP);
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) h[b][q] += src[b * 256 + (sub + 4 * q) * 16 + c16];
        if (lane < KP) { v += src[STAT + lane]; ssum += src[STAT + KP + lane]; }
    }
    acc_to_lds<NB>(h, s_H, lane);
    if (lane < KP) s_s[lane] = ssum;
    wave_sync();
    const double cnt = (double)a.lvl_count[l];
    double *eq = a.eq + (size_t)l * (KP * KP + KP);
    for (int i = lane; i < KP * KP; i += WAVE) {
        const int x = i / KP, y = i % KP;
        eq[i] = (x < K && y < K) ? cnt * a.CCt[i] - s_H[i] : 0.0;
    }
    if (lane < KP) {
        double y = 0.0;
        if (valid) {
            y = a.SC[(size_t)(a.sc_off + l) * KP + lane] + v;
            for (int b = 0; b < K; ++b) y -= a.CCt[lane * KP + b] * s_s[b];
        }
        eq[KP * KP + lane] = y;
    }
}

// Stage 3 (one wave per level, after the cross-rank sum): add the ridge term and solve; write A_i[l].
template <int NB>
__global__ void __launch_bounds__(64) k_level_solve(const double *__restrict__ eq, const int *__restrict__ lvl_count,
                                                    int L, int K, double lambda, double *__restrict__ Arows /*L x KP*/,
                                                    int *__restrict__ fail)
{
    constexpr int KP = Geo<NB>::KP;
    __shared__ double s_A[KP * KP];
    const int l = blockIdx.x, lane = threadIdx.x;
    if (l >= L) return;
    if (lvl_count[l] == 0) return;   // level without samples: the reference never visits it (:147)
    const double *src = eq + (size_t)l * (KP * KP + KP);
    for (int i = lane; i < KP * KP; i += WAVE) s_A[i] = src[i];
    wave_sync();
    if (lane < K) s_A[lane * KP + lane] += lambda;                                      // :174,187
    double b = lane < KP ? src[KP * KP + lane] : 0.0;
    wave_sync();
    const bool ok = chol_solve_lds(s_A, KP, K, b, lane);                                // :175,190
    if (!ok) { if (lane == 0) *fail = 1; return; }
    if (lane < K) Arows[(size_t)l * KP + lane] = b;
}

// ---------------------------------------------------------------------------------------------
// set-up kernels (run once per data set)
// ---------------------------------------------------------------------------------------------
__global__ void k_make_codes(const uint8_t *__restrict__ mtr, const uint8_t *__restrict__ mte, int64_t n, int64_t p,
                             int64_t pitch, uint8_t *__restrict__ codes)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p * pitch) return;
    const int64_t j = t / pitch, i = t % pitch;
    uint8_t cd = CODE_TRAIN;   // pad elements count as train entries with x = 0: never held out
    if (i < n) cd = (mtr[j * n + i] ? CODE_TRAIN : 0) | (mte[j * n + i] ? CODE_TEST : 0);
    codes[t] = cd;
}

// out[c][r] = in[r][c] for an (rows x cols) line-major matrix with pitches; pads keep `padval`
template <typename T>
__global__ void __launch_bounds__(256) k_transpose(const T *__restrict__ in, int64_t rows, int64_t cols, int64_t ipitch,
                                                   T *__restrict__ out, int64_t opitch)
{
    __shared__ T tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = in[r * ipitch + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[c * opitch + r] = tile[tx][k];
    }
}

// per-line sums of squares: yy_train[j] = sum_{train} x^2, yy_all[j] = sum x^2; counts of train / test entries
__global__ void __launch_bounds__(256) k_line_sumsq(const double *__restrict__ vals, const uint8_t *__restrict__ codes,
                                                    int64_t pitch, int len, int lines, double *__restrict__ yy_train,
                                                    double *__restrict__ yy_all, unsigned long long *__restrict__ cnt)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= lines) return;
    double st = 0.0, sa = 0.0;
    unsigned long long ntr = 0, nte = 0;
    for (int i = lane; i < len; i += WAVE) {
        const double x = vals[(size_t)j * pitch + i];
        const int cd = codes[(size_t)j * pitch + i];
        sa += x * x;
        if (cd & CODE_TRAIN) { st += x * x; ++ntr; }
        if (cd & CODE_TEST) ++nte;
    }
    st = wave_sum(st);
    sa = wave_sum(sa);
    for (int o = 32; o >= 1; o >>= 1) { ntr += __shfl_xor(ntr, o, 64); nte += __shfl_xor(nte, o, 64); }
    if (lane == 0) {
        yy_train[j] = st;
        yy_all[j] = sa;
        atomicAdd(&cnt[0], ntr);
        atomicAdd(&cnt[1], nte);
    }
}

// S[j][off_i + l] = sum_{r in level l of covariate i} x_rj    (fixed member order)
__global__ void __launch_bounds__(256) k_level_sums(const double *__restrict__ vals, int64_t pitch, int p,
                                                    const int *__restrict__ members_all /*c x n*/,
                                                    const int *__restrict__ lvl_ptr_all /*SL + c*/,
                                                    const int *__restrict__ lvl_off, int c, int n, int SL, int SLP,
                                                    double *__restrict__ S)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)p * SL) return;
    const int j = (int)(t / SL), gl = (int)(t % SL);
    int i = 0;
    while (i + 1 < c && gl >= lvl_off[i + 1]) ++i;
    const int l = gl - lvl_off[i];
    const int *ptr = lvl_ptr_all + lvl_off[i] + i;   // covariate i's CSR pointer array (L_i + 1 entries)
    const int *mem = members_all + (size_t)i * n;
    double s = 0.0;
    for (int m = ptr[l]; m < ptr[l + 1]; ++m) s += vals[(size_t)j * pitch + mem[m]];
    S[(size_t)j * SLP + gl] = s;
}

// pack / unpack between the host's K x len column-major factor (== len rows of K) and rows of pitch KP
__global__ void k_pack_rows(const double *__restrict__ src /*len x K, row = K contiguous*/, int64_t len, int K, int KP,
                            double *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= len * KP) return;
    const int64_t r = t / KP;
    const int k = (int)(t % KP);
    dst[t] = k < K ? src[r * K + k] : 0.0;
}

__global__ void k_unpack_rows(const double *__restrict__ src, int64_t len, int K, int KP, double *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= len * K) return;
    const int64_t r = t / K;
    const int k = (int)(t % K);
    dst[t] = src[r * KP + k];
}

// A_i host layout is L x K column-major: element (l, k) at l + k L.  Stack rows: Astack[off + l][k].
__global__ void k_pack_A(const double *__restrict__ src, int L, int K, int KP, double *__restrict__ dst_rows)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L * KP) return;
    const int l = t / KP, k = t % KP;
    dst_rows[t] = k < K ? src[l + (size_t)k * L] : 0.0;
}

__global__ void k_unpack_A(const double *__restrict__ src_rows, int L, int K, int KP, double *__restrict__ dst)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L * K) return;
    const int l = t % L, k = t / L;
    dst[t] = src_rows[(size_t)l * KP + k];
}

// out[0..3] = {sum sse_train, sum sse_test, sum c^2, sum |c|}, out[6] = sum of a^2 over all row factors; one
// block, fixed order.  out[4..5] (entry counts) are filled by the host.
__global__ void __launch_bounds__(256) k_loss_reduce(const double *__restrict__ sse_train,
                                                     const double *__restrict__ sse_test, const double *__restrict__ b2,
                                                     const double *__restrict__ b1, int p,
                                                     const double *__restrict__ Astack, int SL, int K, int KP,
                                                     double *__restrict__ out)
{
    __shared__ double red[5][256];
    double s[5] = {0, 0, 0, 0, 0};
    for (int j = threadIdx.x; j < p; j += 256) { s[0] += sse_train[j]; s[1] += sse_test[j]; s[2] += b2[j]; s[3] += b1[j]; }
    for (int i = threadIdx.x; i < SL * KP; i += 256) if (i % KP < K) s[4] += Astack[i] * Astack[i];
    for (int q = 0; q < 5; ++q) red[q][threadIdx.x] = s[q];
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) out[threadIdx.x] = red[threadIdx.x][0];
    if (threadIdx.x == 4) out[6] = red[4][0];
}

// total += sum_j sweeps[j]   (profiling only)
__global__ void __launch_bounds__(256) k_accum_sweeps(const int *__restrict__ sweeps, int p, unsigned long long *total)
{
    __shared__ unsigned long long red[256];
    unsigned long long s = 0;
    for (int j = threadIdx.x; j < p; j += 256) s += (unsigned long long)sweeps[j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total += red[0];
}

// dense X'F over all elements of every line (stand-alone reductions only; optimize() gets it from level sums)
__global__ void __launch_bounds__(64) k_line_dense_xty(const double *__restrict__ vals, int64_t pitch, int len,
                                                       const double *__restrict__ F, int K, int KP,
                                                       double *__restrict__ out)
{
    const int u = blockIdx.x, lane = threadIdx.x;
    for (int k = 0; k < KP; ++k) {
        double s = 0.0;
        if (k < K)
            for (int i = lane; i < len; i += 64) s += vals[(size_t)u * pitch + i] * F[(size_t)i * KP + k];
        s = wave_sum(s);
        if (lane == 0) out[(size_t)u * KP + k] = s;
    }
}

// blocks of sum f~ f~' (stat layout) -> dense K x K column-major XtX = full - complement and Xty = qfull - row K
template <int NB>
__global__ void __launch_bounds__(64) k_stats_to_dense(const double *__restrict__ stat, int nseg, int units, int K,
                                                       const double *__restrict__ full /*KP x KP*/,
                                                       const double *__restrict__ qfull /*units x KP*/,
                                                       double *__restrict__ G_out, double *__restrict__ q_out)
{
    constexpr int KP = Geo<NB>::KP, NBLK = Geo<NB>::NBLK, STAT = Geo<NB>::STAT;
    __shared__ double s_H[KP * KP];
    const int u = blockIdx.x, lane = threadIdx.x;
    const int sub = lane >> 4, c16 = lane & 15;
    d4 h[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) h[b] = d4{0.0, 0.0, 0.0, 0.0};
    for (int sg = 0; sg < nseg; ++sg) {
        const double *src = stat + ((size_t)sg * units + u) * STAT;
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) h[b][q] += src[b * 256 + (sub + 4 * q) * 16 + c16];
    }
    acc_to_lds<NB>(h, s_H, lane);
    wave_sync();
    for (int i = lane; i < K * K; i += WAVE) {
        const int x = i % K, y = i / K;
        G_out[(size_t)u * K * K + i] = full[x * KP + y] - s_H[x * KP + y];
    }
    if (lane < K) q_out[(size_t)u * K + lane] = qfull[(size_t)u * KP + lane] - s_H[K * KP + lane];
}

}  // namespace insider
