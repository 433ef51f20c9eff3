// insider_cont_v2.hpp — optimize_continuous_v2 (src/optimize.cpp:76-137) as a stand-alone operator on an ARBITRARY residual
// matrix `data` (the reference's .Call entry `_insider_optimize_continuous_v2`, src/RcppExports.cpp:69-85), without a handle.
//
// The reference materialises resid = data - z u C and, per coordinate i and pass, two rank-one updates of the n x p residual
// plus a masked n x p reduction (:104-119).  Everything it reads off the residual is a function of two per-gene sums:
//   w_j = sum_{r: M_rj != 0} z_r^2          t_j = sum_{r: M_rj != 0} z_r data_rj
// because  Xty_i (:111) = z' (M % resid_{+i}) c_i' = b_i - sum_{a != i} H_ia u_a   and   XtX_i (:112-114) = H_ii  with
//   H = C diag(w) C'   (K x K),   b = C t   (K)
// so the scalar passes u_i = Xty_i / (XtX_i + lambda) until sum |du| < 0.1 (:117-122) are cyclic coordinate descent on (H, b):
// k_cont_cd, the kernel the resident path uses.  ONE streaming pass over (data, M) — 9 bytes per element, coalesced down
// the gene columns: HBM-bound — then a K x K x p weighted Gram in fixed summation order (bitwise reproducible).
// tuning = 0 (:127-131): Xty = C data' z = C t with t over ALL samples, XtX = (z'z) gram + lambda I from the caller's `gram`,
// solve(..., likely_sympd) = k_level_solve.
#pragma once

namespace insider {

constexpr int CV2_SLAB = 64;   // genes per partial sum of the weighted Gram

// one wave per gene: the two masked sums down the gene's column (MASKED = false: over all samples, w is not needed)
template <bool MASKED>
__global__ void __launch_bounds__(256) k_cv2_gene(const double *__restrict__ data, const uint8_t *__restrict__ ind,
                                                  const double *__restrict__ z, int64_t n, int64_t p, double *__restrict__ w,
                                                  double *__restrict__ t)
{
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= p) return;
    const double *d = data + (size_t)j * n;
    const uint8_t *m = MASKED ? ind + (size_t)j * n : nullptr;
    double sw[4] = {0.0, 0.0, 0.0, 0.0}, st[4] = {0.0, 0.0, 0.0, 0.0};
    int64_t r = lane;
    for (; r + 3 * WAVE < n; r += 4 * WAVE) {   // four loads of each stream in flight per lane
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double zr = z[r + q * WAVE], dv = d[r + q * WAVE];
            const bool on = MASKED ? m[r + q * WAVE] != 0 : true;
            const double zm = on ? zr : 0.0;
            sw[q] = fma(zm, zr, sw[q]);
            st[q] = fma(zm, dv, st[q]);
        }
    }
    for (; r < n; r += WAVE) {
        const double zr = z[r], dv = d[r];
        const bool on = MASKED ? m[r] != 0 : true;
        const double zm = on ? zr : 0.0;
        sw[0] = fma(zm, zr, sw[0]);
        st[0] = fma(zm, dv, st[0]);
    }
    const double tw = wave_sum((sw[0] + sw[1]) + (sw[2] + sw[3])), tt = wave_sum((st[0] + st[1]) + (st[2] + st[3]));
    if (lane == 0) {
        if (MASKED) w[j] = tw;
        t[j] = tt;
    }
}

// z'z in a fixed order (tuning = 0, :128)
__global__ void __launch_bounds__(64) k_cv2_zz(const double *__restrict__ z, int64_t n, double *__restrict__ out)
{
    double acc = 0.0;
    for (int64_t r = threadIdx.x; r < n; r += WAVE) acc = fma(z[r], z[r], acc);
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[0] = acc;
}

// partial (H, b) of one slab of CV2_SLAB genes, in the [KP x KP | KP] layout k_cont_cd / k_level_solve read
// C: K x p column-major (the caller's c_factor); w may be null (tuning = 0: only b is formed)
__global__ void __launch_bounds__(256) k_cv2_eq_part(const double *__restrict__ C, const double *__restrict__ w,
                                                     const double *__restrict__ t, int K, int KP, int64_t p,
                                                     double *__restrict__ part)
{
    __shared__ double cs[CV2_SLAB * 64], ws[CV2_SLAB], ts[CV2_SLAB];
    const int64_t j0 = (int64_t)blockIdx.x * CV2_SLAB;
    const int ng = (int)((p - j0) < CV2_SLAB ? (p - j0) : CV2_SLAB);
    for (int e = threadIdx.x; e < CV2_SLAB * 64; e += 256) {
        const int g = e >> 6, a = e & 63;
        cs[e] = (g < ng && a < K) ? C[(size_t)(j0 + g) * K + a] : 0.0;
    }
    if (threadIdx.x < CV2_SLAB) {
        const bool ok = (int)threadIdx.x < ng;
        ws[threadIdx.x] = (ok && w) ? w[j0 + threadIdx.x] : 0.0;
        ts[threadIdx.x] = ok ? t[j0 + threadIdx.x] : 0.0;
    }
    __syncthreads();
    double *out = part + (size_t)blockIdx.x * (KP * KP + KP);
    for (int o = threadIdx.x; o < KP * KP + KP; o += 256) {
        double acc = 0.0;
        if (o < KP * KP) {
            const int a = o / KP, b = o % KP;
            if (w)
                for (int g = 0; g < CV2_SLAB; ++g) acc = fma(cs[g * 64 + a] * ws[g], cs[g * 64 + b], acc);
        } else {
            const int a = o - KP * KP;
            for (int g = 0; g < CV2_SLAB; ++g) acc = fma(cs[g * 64 + a], ts[g], acc);
        }
        out[o] = acc;
    }
}

// eq = sum of the slab partials in slab order; tuning = 0 (gram != null): the matrix part is (z'z) gram instead (:128)
__global__ void __launch_bounds__(256) k_cv2_eq_sum(const double *__restrict__ part, int nslab, int K, int KP,
                                                    const double *__restrict__ gram /*K x K column-major or null*/,
                                                    const double *__restrict__ zz, double *__restrict__ eq)
{
    const int o = blockIdx.x * 256 + threadIdx.x, len = KP * KP + KP;
    if (o >= len) return;
    double acc = 0.0;
    if (gram && o < KP * KP) {
        const int a = o / KP, b = o % KP;
        acc = (a < K && b < K) ? zz[0] * gram[(size_t)a * K + b] : 0.0;
    } else {
        for (int s = 0; s < nslab; ++s) acc += part[(size_t)s * len + o];
    }
    eq[o] = acc;
}

}  // namespace insider
