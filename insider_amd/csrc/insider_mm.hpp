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

}  // namespace insider
