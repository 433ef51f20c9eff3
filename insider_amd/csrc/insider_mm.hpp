// insider_mm.hpp — the small dense products of the path on v_mfma_f64_16x16x4: one huge dimension (genes, or samples),
// the others <= a few hundred.  They are plain GEMM-shaped work (K x sum(L) x p flops, 3e8 at c3) and HBM-bound on the
// one big operand; the scalar-FMA versions they replace ran at 0.8 TB/s.
//   k_mm_rows:   out[m][n] = sum_k X[m][k] W(k, n)          m huge         (Q = S A, V = C A')
//   k_mm_reduce: part[s][l][n] = sum_{m in slab s} X[m][l] Y[m][n]   m huge   (S'C, U'C, C'C, R'R) + k_sum_partials
// MFMA operand map (A[i][k], B[j][k] with i, j = lane & 15 and k = lane >> 4; D[(lane >> 4) + 4 r][lane & 15]).
#pragma once

namespace insider {

// One wave per 16 rows of X and 16 NT output columns (grid.y tiles the columns); NT accumulator blocks.  WT: W is given transposed (Wt[n][k], row stride ldw),
// else W[k][n].  Kd is padded up to a multiple of 4 by reading zeros (guards).
template <int NT, bool WT>
__global__ void __launch_bounds__(256) k_mm_rows(const double *__restrict__ X, int64_t ldx, int M, int Kd,
                                                 const double *__restrict__ W, int ldw, int N, double *__restrict__ out,
                                                 int64_t ldo, int n_store)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = (blockIdx.x * 4 + w) * 16;
    if (m0 >= M) return;
    const int g = lane >> 4, c16 = lane & 15;
    const int row = m0 + c16 < M ? m0 + c16 : M - 1;
    const int nb = blockIdx.y * 16 * NT;   // this block's first output column
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    const double *xr = X + (size_t)row * ldx;
    for (int k0 = 0; k0 < Kd; k0 += 16) {   // four MFMA steps per trip: all their loads are issued before the first MFMA
        double a[4], b[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 4 * s + g;
            const bool kin = k < Kd;
            a[s] = kin ? xr[k] : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = nb + 16 * t + c16;
                b[s][t] = 0.0;
                if (kin && n < N) b[s][t] = WT ? W[(size_t)n * ldw + k] : W[(size_t)k * ldw + n];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s][t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + g + 4 * r, n = nb + 16 * t + c16;
            if (m < M && n < n_store) out[(size_t)m * ldo + n] = n < N ? acc[t][r] : 0.0;
        }
}

// grid = (slabs, ceil(L / 16)); one wave per (slab of `per` rows, 16 columns of X); NT = ceil(N / 16) blocks.
// part[(slab * L + l) * ldo + n]
template <int NT>
__global__ void __launch_bounds__(64) k_mm_reduce(const double *__restrict__ X, int64_t ldx, const double *__restrict__ Y,
                                                  int64_t ldy, int M, int per, int L, int N, double *__restrict__ part,
                                                  int ldo)
{
    const int lane = threadIdx.x;
    const int g = lane >> 4, c16 = lane & 15;
    const int m_begin = blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
    const int l = blockIdx.y * 16 + c16;
    const bool lin = l < L;
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    for (int m0 = m_begin; m0 < m_end; m0 += 16) {   // four MFMA steps per trip, loads first
        double a[4], b[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + 4 * s + g;
            const bool min_ = m < m_end;
            const size_t mr = (size_t)(min_ ? m : m_end - 1);
            a[s] = (min_ && lin) ? X[mr * ldx + l] : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = 16 * t + c16;
                b[s][t] = (min_ && n < N) ? Y[mr * ldy + n] : 0.0;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s][t], acc[t], 0, 0, 0);
    }
    double *out = part + (size_t)blockIdx.x * L * ldo;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lo = blockIdx.y * 16 + g + 4 * r, n = 16 * t + c16;
            if (lo < L && n < ldo) out[(size_t)lo * ldo + n] = n < N ? acc[t][r] : 0.0;
        }
}

// ---- round 5: the same two products with their operands moved once ------------------------------------------------------------
// k_mm_rows spends most of its instructions re-reading W: every wave fetches the whole W tile (Kd x 16 NT doubles) from the
// cache hierarchy for its 16 rows of X, and reads X in 8-byte pieces (32 contiguous bytes per row and instruction): 1.0 - 1.6 TB/s
// at c3 on data that is streamed once.  k_mm_rows2: W is staged once per block in LDS in MFMA operand order (one conflict-free
// ds_read_b64 per instruction), a lane reads its row of X in 16-byte pieces — four consecutive k per lane and chunk of 16, so the
// k of MFMA step e of chunk S are 16 S + 4 g + e (g = lane >> 4): the staged W follows the same map — all loads of up to eight
// chunks are issued before the first MFMA, and a wave takes tiles_per_wave consecutive tiles of 16 rows.  Sums over k in
// another order than k_mm_rows: results agree to rounding.  X rows must be 16-byte aligned (X, ldx and kread even).
template <int NT, bool WT>
__global__ void __launch_bounds__(256) k_mm_rows2(const double *__restrict__ X, int64_t ldx, int M, int Kd,
                                                  const double *__restrict__ W, int ldw, int N, double *__restrict__ out,
                                                  int64_t ldo, int n_store, int tiles_per_wave, int kread, int accumulate)
{
    // kread: columns that may be read from a row of X as given (ldx, or less when X points into a row: a column window of the
    // product); accumulate: out += (the window's product) instead of out =
    extern __shared__ double s_w[];   // [4 nchunk][NT][64]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, c16 = lane & 15;
    const int nchunk = (Kd + 15) >> 4;
    const int nb = blockIdx.y * 16 * NT;   // this block's first output column
    for (int i = threadIdx.x; i < 4 * nchunk * NT * 64; i += 256) {
        const int ln = i & 63, t = (i >> 6) % NT, ks = i / (64 * NT);
        const int k = 16 * (ks >> 2) + 4 * (ln >> 4) + (ks & 3), n = nb + 16 * t + (ln & 15);
        s_w[i] = (k < Kd && n < N) ? (WT ? W[(size_t)n * ldw + k] : W[(size_t)k * ldw + n]) : 0.0;
    }
    __syncthreads();
    const int ntiles = (M + 15) >> 4;
    const int tile0 = (blockIdx.x * 4 + w) * tiles_per_wave;
    const double *wl = s_w + lane;
    for (int tile = tile0; tile < tile0 + tiles_per_wave && tile < ntiles; ++tile) {
        const int m0 = tile * 16;
        const int row = m0 + c16 < M ? m0 + c16 : M - 1;
        const double *xr = X + (size_t)row * ldx + 4 * g;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        if (accumulate) {   // wave-uniform
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + g + 4 * r, n = nb + 16 * t + c16;
                    if (m < M && n < N) acc[t][r] = out[(size_t)m * ldo + n];
                }
        }
        for (int S0 = 0; S0 < nchunk; S0 += 8) {
            double x[8][4];
#pragma unroll
            for (int S = 0; S < 8; ++S)
                if (S0 + S < nchunk) {   // wave-uniform
                    const int k4 = 16 * (S0 + S) + 4 * g;
                    double2 lo = {0.0, 0.0}, hi = {0.0, 0.0};
                    if (k4 + 1 < kread) lo = *reinterpret_cast<const double2 *>(xr + 16 * (S0 + S));
                    if (k4 + 3 < kread) hi = *reinterpret_cast<const double2 *>(xr + 16 * (S0 + S) + 2);
                    x[S][0] = lo.x; x[S][1] = lo.y; x[S][2] = hi.x; x[S][3] = hi.y;
                    if (16 * (S0 + S) + 16 > Kd) {   // the last chunk: what lies beyond Kd in the row is not part of the product
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[S][e] = k4 + e < Kd ? x[S][e] : 0.0;
                    }
                }
#pragma unroll
            for (int S = 0; S < 8; ++S)
                if (S0 + S < nchunk) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[S][e], wl[((size_t)(4 * (S0 + S) + e) * NT + t) * 64], acc[t], 0, 0, 0);
                }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + g + 4 * r, n = nb + 16 * t + c16;
                if (m < M && n < n_store) out[(size_t)m * ldo + n] = n < N ? acc[t][r] : 0.0;
            }
    }
}

// k_mm_reduce with LT tiles of 16 columns of X per wave (Y is read once per LT tiles instead of once per tile) and the operands
// of the next TWO trips of 16 rows in flight during the MFMAs of the current one.  Every output entry adds the same products in
// the same order as in k_mm_reduce: bit-identical partial sums.  grid = (slabs, ceil(ceil(L / 16) / LT)).
template <int NT, int LT>
__global__ void __launch_bounds__(64) k_mm_reduce2(const double *__restrict__ X, int64_t ldx, const double *__restrict__ Y,
                                                   int64_t ldy, int M, int per, int L, int N, double *__restrict__ part,
                                                   int ldo)
{
    const int lane = threadIdx.x;
    const int g = lane >> 4, c16 = lane & 15;
    const int m_begin = blockIdx.x * per, m_end = m_begin + per < M ? m_begin + per : M;
    const int lt0 = blockIdx.y * LT;
    const int ltn = (L + 15) >> 4;
    d4 acc[LT][NT];
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[l][t] = d4{0.0, 0.0, 0.0, 0.0};
    struct Ops { double a[4][LT], b[4][NT]; };
    auto fetch = [&](int m0, Ops &o) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + 4 * s + g;
            const bool min_ = m < m_end;
            const size_t mr = (size_t)(min_ ? m : m_end - 1);
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                const int lc = 16 * (lt0 + l) + c16;
                o.a[s][l] = (min_ && lc < L) ? X[mr * ldx + lc] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = 16 * t + c16;
                o.b[s][t] = (min_ && n < N) ? Y[mr * ldy + n] : 0.0;
            }
        }
    };
    auto compute = [&](const Ops &o) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int l = 0; l < LT; ++l)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[l][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.a[s][l], o.b[s][t], acc[l][t], 0, 0, 0);
    };
    Ops o0, o1, o2;
    if (m_begin < m_end) fetch(m_begin, o0);
    if (m_begin + 16 < m_end) fetch(m_begin + 16, o1);
    for (int m0 = m_begin; m0 < m_end; m0 += 48) {
        if (m0 + 32 < m_end) fetch(m0 + 32, o2);
        compute(o0);
        if (m0 + 16 < m_end) {
            if (m0 + 48 < m_end) fetch(m0 + 48, o0);
            compute(o1);
            if (m0 + 32 < m_end) {
                if (m0 + 64 < m_end) fetch(m0 + 64, o1);
                compute(o2);
            }
        }
    }
    double *out = part + (size_t)blockIdx.x * L * ldo;
#pragma unroll
    for (int l = 0; l < LT; ++l)
        if (lt0 + l < ltn) {   // wave-uniform
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lo = 16 * (lt0 + l) + g + 4 * r, n = 16 * t + c16;
                    if (lo < L && n < ldo) out[(size_t)lo * ldo + n] = n < N ? acc[l][t][r] : 0.0;
                }
        }
}

}  // namespace insider
