// insider_row_merged.hpp — the masked row update (optimize_row, src/optimize.cpp:150-176) without per-sample
// statistics: categorical covariates only.
//
// For covariate i, level l, H(j) the held-out samples of gene j and s_r = sum_{m != i} A_m[level_m(r)]:
//   XtX_l = sum_{r in l} (CC' - Hc_r)            = |l| CC' - sum_j m_lj c_j c_j',      m_lj = |l ∩ H(j)|
//   Xty_l = sum_{r in l} sum_{j train} c_j (x_rj - c_j's_r)
//         = (S_i^train C')[l] - CC' sum_{r in l} s_r + sum_j u_j[l] c_j,
//     u_j[l] = sum_{r in l ∩ H(j)} c_j's_r = sum_{r in l ∩ H(j)} sum_{m != i} (A_m c_j)[level_m(r)].
// The rank-one term c_j c_j' is shared by all held-out samples of gene j inside a level, so the K x K work is one
// WEIGHTED rank-one update per (level, gene) pair — 5.5e6 pairs against 5e7 held-out entries at c3 — and every held-out
// entry costs one table look-up and one add.  Everything that depends on the data only is built once per data set:
// the genes' held-out samples grouped by level (per covariate), the (level -> gene, count) lists, the train-only
// per-level sums S^train, and the level-pair sample counts that give sum_{r in l} s_r.  All terms stay sums over genes,
// so gene-axis sharding reduces them with the same all-reduce as before.
#pragma once

namespace insider {

// ---- once per data set -------------------------------------------------------------------------------------------
// grp[j][l] (absolute positions into the gene's slice of the sorted-entry array, which reuses the column list's
// offsets): start of the held-out samples of gene j that fall into level l of this covariate; grp[j][L] = end.
__global__ void __launch_bounds__(256) k_group_count(const uint32_t *__restrict__ col_ptr, const int *__restrict__ col_idx,
                                                     const int *__restrict__ lev /*n*/, int L, int p,
                                                     uint32_t *__restrict__ grp /*[p][L + 1]*/)
{
    extern __shared__ int s_hist[];   // [4][L]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= p) return;
    int *hist = s_hist + w * L;
    for (int l = lane; l < L; l += WAVE) hist[l] = 0;
    wave_sync();
    for (uint32_t e = col_ptr[j] + lane; e < col_ptr[j + 1]; e += WAVE) {
        const int r = col_idx[e];
        if (r != LIST_PAD) atomicAdd(&hist[lev[r]], 1);   // integer counts: order-independent
    }
    wave_sync();
    // exclusive scan: each lane owns a contiguous run of levels
    const int seg = (L + WAVE - 1) / WAVE, l0 = lane * seg, l1 = l0 + seg < L ? l0 + seg : L;
    int run = 0;
    for (int l = l0; l < l1; ++l) run += hist[l];
    int incl = run;
    for (int o = 1; o < WAVE; o <<= 1) {
        const int t = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl += t;
    }
    uint32_t pos = col_ptr[j] + (uint32_t)(incl - run);
    uint32_t *g = grp + (size_t)j * (L + 1);
    for (int l = l0; l < l1; ++l) { g[l] = pos; pos += (uint32_t)hist[l]; }
    if (lane == WAVE - 1) g[L] = col_ptr[j] + (uint32_t)incl;
}

// For every gene, its held-out samples stably grouped by this covariate's level (ascending sample id inside a group);
// what is kept per entry is not the sample id but the STACKED level index of each OTHER covariate (uint16, one
// plane of `entries` values per other covariate, in covariate order): all the merged update needs from an entry.
__global__ void __launch_bounds__(256) k_group_fill(const uint32_t *__restrict__ col_ptr, const int *__restrict__ col_idx,
                                                    const int *__restrict__ lev_all /*c x n*/, const int *__restrict__ lvl_off,
                                                    int c, int n, int cov, int L, int p, const uint32_t *__restrict__ grp,
                                                    uint16_t *__restrict__ slev, size_t plane)
{
    extern __shared__ int s_hist[];   // [4][L]: running cursor per level
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= p) return;
    int *cur = s_hist + w * L;
    const uint32_t *g = grp + (size_t)j * (L + 1);
    const int *lev = lev_all + (size_t)cov * n;
    for (int l = lane; l < L; l += WAVE) cur[l] = (int)g[l];
    wave_sync();
    const uint64_t lt = lanemask_lt(lane);
    for (uint32_t base = col_ptr[j]; base < col_ptr[j + 1]; base += WAVE) {
        const uint32_t e = base + lane;
        const int r = e < col_ptr[j + 1] ? col_idx[e] : LIST_PAD;
        const int l = r != LIST_PAD ? lev[r] : -1;
        uint64_t todo = __ballot(l >= 0);
        while (todo) {                                    // one round per distinct level among the 64 entries
            const int lead = __ffsll((unsigned long long)todo) - 1;
            const int ll = __builtin_amdgcn_readlane(l, lead);
            const uint64_t same = __ballot(l == ll);
            const int c0 = cur[ll];
            if (l == ll) {
                const size_t pos = (size_t)c0 + __popcll(same & lt);
                int o = 0;
                for (int m = 0; m < c; ++m)
                    if (m != cov) slev[(size_t)(o++) * plane + pos] = (uint16_t)(lvl_off[m] + lev_all[(size_t)m * n + r]);
            }
            wave_sync();
            if (lane == lead) cur[ll] = c0 + __popcll(same);
            wave_sync();
            todo &= ~same;
        }
    }
}

// ---- per outer iteration and covariate -----------------------------------------------------------------------------
// V[j][q] = (A c_j)[q] for every stacked level q is the small product C A' (k_mm_rows, insider_mm.hpp); after a
// covariate's update only its own columns are recomputed.

// U[j][l] = sum over the held-out samples r of gene j in level l of sum_{m != cov} (A_m c_j)[level_m(r)].  One wave
// per gene: its row of V goes to LDS; then, 64 levels at a time (their held-out samples are one contiguous range of
// the level-sorted list), lanes walk the range entry-wise in tiles (coalesced uint16 reads), leave each entry's
// value in LDS, and lane l adds up its own group in list order (fixed order: bitwise reproducible).
constexpr int GU_TILE = 512;    // entries per tile (LDS doubles per wave)

template <int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_gene_u(const uint32_t *__restrict__ grp, const uint16_t *__restrict__ slev, size_t plane, int nother, int L, int LP,
         const double *__restrict__ V, int SLP, int p, int SL, double *__restrict__ U)
{
    extern __shared__ double s_gu[];   // [WPB][SL + GU_TILE]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * WPB + w;
    if (j >= p) return;
    double *v = s_gu + (size_t)w * (SL + GU_TILE), *val = v + SL;
    for (int q = lane; q < SL; q += WAVE) v[q] = V[(size_t)j * SLP + q];
    wave_sync();
    const uint32_t *g = grp + (size_t)j * (L + 1);
    for (int l0 = 0; l0 < LP; l0 += WAVE) {
        // nl levels in this pass, T lanes per level (T a power of two, the same for every gene: fixed summation order)
        const int lhi = l0 + WAVE < L ? l0 + WAVE : L;
        const int nl = lhi > l0 ? lhi - l0 : 1;
        int T = 1;
        while (2 * T * nl <= WAVE) T *= 2;
        const int lvl = lane / T, sub = lane % T, l = l0 + lvl;
        const bool mine = lvl < nl && l < L;
        const uint32_t gb = mine ? g[l] : 0, ge = mine ? g[l + 1] : 0;          // this lane's group
        const uint32_t r0 = g[l0 < L ? l0 : L], r1 = g[lhi > l0 ? lhi : L];      // the pass's range (wave-uniform)
        double acc = 0.0;
        for (uint32_t t0 = r0; t0 < r1; t0 += GU_TILE) {
            const uint32_t t1 = t0 + GU_TILE < r1 ? t0 + GU_TILE : r1;
            for (uint32_t t = t0 + lane; t < t1; t += 4 * WAVE) {   // four entries per lane in flight
                double x[4] = {0.0, 0.0, 0.0, 0.0};
                for (int o = 0; o < nother; ++o) {
                    int q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t tu = t + u * WAVE;
                        q[u] = slev[(size_t)o * plane + (tu < t1 ? tu : t1 - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] += v[q[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t tu = t + u * WAVE;
                    if (tu < t1) val[tu - t0] = x[u];
                }
            }
            wave_sync();
            const uint32_t b = gb > t0 ? gb : t0, e = ge < t1 ? ge : t1;
            // sub-lane `sub` takes entries b + sub, b + sub + T, ... of the group (positions relative to the GROUP start,
            // so that the split does not depend on the tiling)
            uint32_t t = b + ((sub + T - ((b - gb) % T)) % T);
            for (; t < e; t += T) acc += val[t - t0];
            wave_sync();
        }
        // the T partial sums of a level, added in sub-lane order
        val[lane] = acc;
        wave_sync();
        if (sub == 0 && mine) {
            double tot = 0.0;
            for (int q = 0; q < T; ++q) tot += val[lane + q];
            U[(size_t)j * LP + l] = tot;
        }
        if (lane == 0)
            for (int lz = L > l0 ? L : l0; lz < l0 + WAVE && lz < LP; ++lz) U[(size_t)j * LP + lz] = 0.0;   // pitch padding
        wave_sync();
    }
}

// weighted SYRK over (gene, weight) lists: stat[item] = lower 16x16 blocks of sum_e w_e c_e c_e' for the list range
// [item_begin, item_end) (multiples of LIST_ALIGN; padding = (LIST_PAD, 0.0)).  Same pipeline as k_list_stats; the
// weight scales the A operand (NB extra multiplies per four entries), the value slot of the row is not used.
template <int NB>
__device__ __forceinline__ void wsyrk_load(const int *li, const double *lw, int batch, __amdgpu_buffer_rsrc_t rsrc,
                                           double (&a)[SYRK_GB][NB], double (&aw)[SYRK_GB][NB], int lane)
{
    constexpr int RB = Geo<NB>::KP * 8;
    const int sub = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int u = 0; u < SYRK_GB; ++u) {
        const int idx = li[SYRK_BATCH * batch + 4 * u + sub];
        const double wv = lw[SYRK_BATCH * batch + 4 * u + sub];
        const int off = idx * RB + c16 * 8;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const v2i x = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 128 * b, 0);
            a[u][b] = __hiloint2double(x.y, x.x);
            aw[u][b] = a[u][b] * wv;
        }
    }
}

template <int NB>
__device__ __forceinline__ void wsyrk_mfma(const double (&a)[SYRK_GB][NB], const double (&aw)[SYRK_GB][NB],
                                           d4 (&acc)[Geo<NB>::NBLK])
{
#pragma unroll
    for (int u = 0; u < SYRK_GB; ++u) {
        int blk = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj, ++blk)
                acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[u][bi], a[u][bj], acc[blk], 0, 0, 0);
    }
}

template <int NB, int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_wsyrk(const uint32_t *__restrict__ item_begin, const uint32_t *__restrict__ item_end, int nitems,
        const int *__restrict__ lidx, const double *__restrict__ lw, const double *__restrict__ F, int64_t f_rows,
        double *__restrict__ stat)
{
    constexpr int NBLK = Geo<NB>::NBLK;
    __shared__ int s_li[WPB][2][LIST_BLOCK];
    __shared__ double s_lw[WPB][2][LIST_BLOCK];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * WPB + w;
    if (item >= nitems) return;
    d4 acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    const uint32_t first = item_begin[item], last = item_end[item];
    if (first < last) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(F), 0, (int)(f_rows * Geo<NB>::KP * 8), 0x00020000);
        const int nblk = __builtin_amdgcn_readfirstlane((int)((last - first + LIST_BLOCK - 1) / LIST_BLOCK));
        const int nbt = __builtin_amdgcn_readfirstlane((int)((last - first) / SYRK_BATCH));   // even
        constexpr int EPL = LIST_BLOCK / WAVE;
        int ia[EPL], ib[EPL];
        double xa[EPL], xb[EPL];
        auto ld = [&](int blk, int (&ii)[EPL], double (&xx)[EPL]) {
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                const uint32_t e = first + (uint32_t)blk * LIST_BLOCK + (uint32_t)(t * WAVE + lane);
                const uint32_t ec = e < last ? e : last - 1;
                const int iv = lidx[ec];
                const double xv = lw[ec];
                ii[t] = e < last ? iv : LIST_PAD;
                xx[t] = e < last ? xv : 0.0;
            }
        };
        ld(0, ia, xa);
        ld(1, ib, xb);
        for (int blk = 0; blk < nblk; ++blk) {
            int *li = s_li[w][blk & 1];
            double *lx = s_lw[w][blk & 1];
#pragma unroll
            for (int t = 0; t < EPL; ++t) { li[t * WAVE + lane] = ia[t]; lx[t * WAVE + lane] = xa[t]; }
#pragma unroll
            for (int t = 0; t < EPL; ++t) { ia[t] = ib[t]; xa[t] = xb[t]; }
            ld(blk + 2, ib, xb);
            wave_sync();
            const int left = nbt - blk * (LIST_BLOCK / SYRK_BATCH);
            const int nbatch = __builtin_amdgcn_readfirstlane(left < LIST_BLOCK / SYRK_BATCH ? left : LIST_BLOCK / SYRK_BATCH);
            double a0[SYRK_GB][NB], w0[SYRK_GB][NB], a1[SYRK_GB][NB], w1[SYRK_GB][NB];
            wsyrk_load<NB>(li, lx, 0, rsrc, a0, w0, lane);
            for (int b = 0; b < nbatch; b += 2) {
                wsyrk_load<NB>(li, lx, b + 1, rsrc, a1, w1, lane);
                wsyrk_mfma<NB>(a0, w0, acc);
                wsyrk_load<NB>(li, lx, b + 2 < nbatch ? b + 2 : nbatch - 1, rsrc, a0, w0, lane);
                wsyrk_mfma<NB>(a1, w1, acc);
            }
        }
    }
    double *out = stat + (size_t)item * Geo<NB>::STAT;
    const int sub = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[b * 256 + (sub + 4 * r) * 16 + c16] = acc[b][r];
}

// ---- the same weighted SYRK as ONE GEMM over genes (round 3; covariates with many levels) --------------------------------
//   G_l[a][b] = sum_j n_jl c_ja c_jb = (N' P)[l][(a, b)],   N = [genes x levels] held-out counts, P = [genes x pairs], P_j(a, b) = c_ja c_jb
// with the pairs a >= b packed densely (K (K + 1) / 2 of them, 16 per MFMA tile).  k_wsyrk spends NB (NB + 1) / 2 MFMAs on
// every (level, four genes) — full padded 16 x 16 blocks — i.e. L x 3 per four genes at KP = 32; the GEMM needs
// ceil(L / 16) x ceil(pairs / 16): 7 x 30 = 210 against 300 at c3 (K = 30, L = 100), 7 x 14 against 300 at K = 20.  And the
// vector work around the MFMAs (which costs MFMA issue time in an f64 kernel) shrinks: per step of four
// genes a wave converts LT counts and forms two products for 2 LT MFMAs, where k_wsyrk does ~8 instructions per 3 MFMAs.
// The counts come as 1/2 n from the pair-count kernel's static float table (ColFacArgs::hn; the sum is doubled at the end).
// Block = 4 waves = 4 x 2 pair tiles on the SAME gene slab (their count loads hit the same lines); grid.y = gene slabs,
// grid.z = chunks of LT level tiles.  Padded genes (the table and C have four zero rows after the last gene) and padded pairs
// (column KP - 1 of C, always zero) contribute exact zeros: no masks.  Partial sums per slab, added in slab order by
// k_wgemm_sum (fixed order: reproducible), which also doubles them and writes the level records' blocks.
template <int LT>
__global__ void __launch_bounds__(256)
k_wgemm(const float *__restrict__ hn, int hn_stride, int lt_total, const double *__restrict__ C, int KP, int p, int slab, int nslab,
        const uint8_t *__restrict__ pair_ab /*[2][16 ntile]*/, int ntile, double *__restrict__ part /*[slabs][16 lt_total][16 ntile]*/)
{
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // work items = (gene slab, pair of pair tiles), dealt to the waves of the grid densely (round 4: the grid used to carry a
    // mostly empty fourth block per slab — 276 blocks for 256 CUs, i.e. a second round of 20 blocks that doubled the kernel's
    // time; now nslab x ceil(ntile / 2) waves, sized by the host to at most one wave per SIMD)
    const int per = (ntile + 1) >> 1;
    const int W = blockIdx.x * 4 + w;
    if (W >= per * nslab) return;
    const int slab_id = W / per;
    const int T0 = (W - slab_id * per) * 2;              // this wave's pair tiles T0, T0 + 1
    const int lt0 = blockIdx.z * LT;                     // its level tiles lt0 .. lt0 + LT - 1 (beyond lt_total: skipped at the end)
    const int g4 = lane >> 4, c16 = lane & 15;
    const int j_begin = slab_id * slab, j_end = j_begin + slab < p ? j_begin + slab : p;
    // lane offsets in BYTES, fixed for the whole kernel: every load is a wave-uniform base + one of these (+ a constant)
    unsigned off_a[2], off_b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int T = T0 + t < ntile ? T0 + t : ntile - 1;
        off_a[t] = 8u * (unsigned)(g4 * KP + pair_ab[16 * T + c16]);
        off_b[t] = 8u * (unsigned)(g4 * KP + pair_ab[16 * ntile + 16 * T + c16]);
    }
    const unsigned off_n = 4u * (unsigned)(g4 * hn_stride + (c16 & 3) * 4 + (c16 >> 2));   // level c16 of a tile, gene g4 of a step
    d4 acc[LT][2];
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[l][t] = d4{0.0, 0.0, 0.0, 0.0};
    struct Ops { float n[LT]; double ca[2], cb[2]; };
    auto fetch = [&](int j0, Ops &o) {
        const char *hb = reinterpret_cast<const char *>(hn + (size_t)j0 * hn_stride + 16 * lt0);
        const char *cb = reinterpret_cast<const char *>(C + (size_t)j0 * KP);
#pragma unroll
        for (int l = 0; l < LT; ++l) o.n[l] = *reinterpret_cast<const float *>(hb + off_n + 64 * l);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            o.ca[t] = *reinterpret_cast<const double *>(cb + off_a[t]);
            o.cb[t] = *reinterpret_cast<const double *>(cb + off_b[t]);
        }
    };
    auto compute = [&](const Ops &o) {
        double b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = o.ca[t] * o.cb[t];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const double a = (double)o.n[l];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[l][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t], acc[l][t], 0, 0, 0);
        }
    };
    // three operand sets: the loads of the next TWO steps of four genes are in flight during the MFMAs of the current one (the
    // kernel shares the machine with the main stream's memory-bound chain: one step of MFMAs does not cover a load's latency)
    Ops o0, o1, o2;
    if (j_begin < j_end) fetch(j_begin, o0);
    if (j_begin + 4 < j_end) fetch(j_begin + 4, o1);
    for (int j0 = j_begin; j0 < j_end; j0 += 12) {
        if (j0 + 8 < j_end) fetch(j0 + 8, o2);
        compute(o0);
        if (j0 + 4 < j_end) {
            if (j0 + 12 < j_end) fetch(j0 + 12, o0);
            compute(o1);
            if (j0 + 8 < j_end) {
                if (j0 + 16 < j_end) fetch(j0 + 16, o1);
                compute(o2);
            }
        }
    }
    double *out = part + (size_t)slab_id * (16 * lt_total) * (16 * ntile);
#pragma unroll
    for (int l = 0; l < LT; ++l)
        if (lt0 + l < lt_total) {   // wave-uniform
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (T0 + t < ntile) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[(size_t)(16 * (lt0 + l) + g4 + 4 * r) * (16 * ntile) + 16 * (T0 + t) + c16] = acc[l][t][r];
                }
        }
}

// (Round 5 built this GEMM on v_mfma_f64_4x4x4 as well — k_wgemm4, commit 5e6b267: count operand as the A operand, the pair
// products rotated by 4 x lanes for the twelve off-diagonal 4 x 4 tiles — and it ran 141 us (rotations by DPP) / 190 us (rotated
// products formed from C: 23 gathers per step) where this kernel takes 128 beside k_gene_u_cnt: the segment is bound by what the
// two kernels need together, not by this one's matrix rate.  profiles/r05/exp/ab_wgemm_4x4x4_*.log.)
// rec[l][e] for every entry e of the level record's lower 16 x 16 blocks: 2 sum_slabs part[slab][l][pair(a, b)] inside the
// K x K part (slab order; pair(a, b) = hi (hi + 1) / 2 + lo, the packing of the table k_wgemm reads), 0 in the padding
__global__ void __launch_bounds__(256) k_wgemm_sum(const double *__restrict__ part, int nslab, int lt_total, int ntile, int K,
                                                   int stat_len, double *__restrict__ rec, int plen)
{
    const int e = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (e >= stat_len) return;
    int blk = e >> 8, bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
    const int bj = blk - bi * (bi + 1) / 2;
    const int a = 16 * bi + ((e >> 4) & 15), b = 16 * bj + (e & 15);
    double s = 0.0;
    if (a < K && b < K) {
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const size_t stride = (size_t)(16 * lt_total) * (16 * ntile);
        const double *src = part + (size_t)l * (16 * ntile) + (hi * (hi + 1) / 2 + lo);
#pragma unroll 8
        for (int sl = 0; sl < nslab; ++sl) s += src[(size_t)sl * stride];   // (loads batched by the unrolling, sums in slab order)
        s *= 2.0;
    }
    rec[(size_t)l * plen + e] = s;
}

// the tail of a level's summed-partials record (see k_level_reduce): v = (U'C)[l], sum_{r in l} s_r from the
// level-pair sample counts, and |l|.  Block = one level: 4 strided groups of stacked levels x 64 coordinates, the four
// partial sums added in group order.
__global__ void __launch_bounds__(256) k_level_pack(const double *__restrict__ Y /*[L][KP]*/,
                                                    const double *__restrict__ paircnt /*[L][SL]*/, int SL,
                                                    const double *__restrict__ Astack, const int *__restrict__ lvl_count,
                                                    int L, int K, int KP, int stat_len, double *__restrict__ rec)
{
    __shared__ double red[4][64];
    const int l = blockIdx.x, k = threadIdx.x & 63, g = threadIdx.x >> 6;
    if (l >= L) return;
    double ss = 0.0;
    if (k < K)
        for (int q = g; q < SL; q += 4) ss = fma(paircnt[(size_t)l * SL + q], Astack[(size_t)q * KP + k], ss);
    red[g][k] = ss;
    __syncthreads();
    if (g != 0) return;
    double *out = rec + (size_t)l * (stat_len + 2 * KP + 2) + stat_len;
    if (k < KP) {
        out[k] = k < K ? Y[(size_t)l * KP + k] : 0.0;
        out[KP + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
    if (k == 0) { out[2 * KP] = (double)lvl_count[l]; out[2 * KP + 1] = 0.0; }
}

// level_pack + level_reduce (+ level_solve) of the merged update in ONE launch, one block per level: the three kernels are
// 10 - 15 us each at c3 for 6 KB of work per level — launch and drain latency on the serial chain of the row phase.  Waves
// 0..3 form sum_{r in l} s_r from the level-pair sample counts (k_level_pack's sum, same order); wave 0 then builds the
// level's normal equations exactly as k_level_reduce does (eq is still written: the gene-sharded path all-reduces it) and,
// with do_solve, solves them as k_level_solve does — same arithmetic on the same values, bit-identical results.
// Y: (U'C)[l], or — ypart_n > 0 — its per-slab partial sums part[slab][L][KP] straight from k_mm_reduce: the block then adds
// them up itself, in k_sum_partials' order (16 strided groups, then the groups in order: the same bits), which takes that
// kernel and its launch off the main chain of the row phase.
template <int NB>
__global__ void __launch_bounds__(256) k_level_merged(const double *__restrict__ rec /*[L][STAT + 2 KP + 2]: weighted-SYRK level sums*/,
                                                      const double *__restrict__ Y /*[L][KP]*/, int ypart_n,
                                                      const double *__restrict__ paircnt /*[L][SL]*/, int SL,
                                                      const double *__restrict__ Astack, const int *__restrict__ lvl_count,
                                                      const double *__restrict__ CCt, const double *__restrict__ SC /*rows of this covariate*/,
                                                      int L, int K, double lambda, int do_solve, double *__restrict__ eq,
                                                      double *__restrict__ Arows /*L x KP*/, int *__restrict__ fail,
                                                      const double *__restrict__ cnt_real = nullptr /*[L]: real-valued |l| (sum_r z_r^2 of a continuous column) instead of lvl_count*/)
{
    constexpr int KP = Geo<NB>::KP, NBLK = Geo<NB>::NBLK, STAT = Geo<NB>::STAT;
    __shared__ double red[4][64];
    __shared__ double s_H[KP * KP];
    __shared__ double s_A[NB <= 2 ? KP * KP : 1];   // the in-kernel solve is the register route (K <= 31); larger K: k_level_solve
    __shared__ double s_s[KP];
    __shared__ double s_y[16][KP];
    const int l = blockIdx.x, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    if (l >= L) return;
    if (ypart_n > 0) {   // wave-uniform.  k_sum_partials: s_m = sum_{b = m, m + 16, ...} part[b][o], then sum_m s_m in order
        const size_t len = (size_t)L * KP;
        for (int c = threadIdx.x; c < 16 * KP; c += 256) {
            const int m = c / KP, o = c % KP;
            double sm = 0.0;
#pragma unroll 8
            for (int b = m; b < ypart_n; b += 16) sm += Y[(size_t)b * len + (size_t)l * KP + o];
            s_y[m][o] = sm;
        }
    }
    double ss = 0.0;
    if (lane < K)
        for (int q = g; q < SL; q += 4) ss = fma(paircnt[(size_t)l * SL + q], Astack[(size_t)q * KP + lane], ss);
    red[g][lane] = ss;
    __syncthreads();
    if (g != 0) return;
    const bool valid = lane < K;
    const int sub = lane >> 4, c16 = lane & 15;
    const double cnt = cnt_real ? cnt_real[l] : (double)lvl_count[l];
    double v = 0.0;
    if (valid) {
        if (ypart_n > 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) v += s_y[m][lane];
        } else {
            v = Y[(size_t)l * KP + lane];
        }
    }
    const double ssum = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const double *src = rec + (size_t)l * (STAT + 2 * KP + 2);
    d4 h[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) h[b][q] = src[b * 256 + (sub + 4 * q) * 16 + c16];
    acc_to_lds<NB>(h, s_H, lane);
    if (lane < KP) s_s[lane] = ssum;
    wave_sync();
    double *eql = eq + (size_t)l * (KP * KP + KP);
    for (int i = lane; i < KP * KP; i += WAVE) {
        const int x = i / KP, y = i % KP;
        const double e = (x < K && y < K) ? cnt * CCt[i] - s_H[i] : 0.0;
        eql[i] = e;
        if constexpr (NB <= 2) s_A[i] = e;
    }
    double yv = 0.0;
    if (lane < KP) {
        if (valid) {
            yv = SC[(size_t)l * KP + lane] + v;
            for (int b = 0; b < K; ++b) yv -= CCt[b * KP + lane] * s_s[b];      // CC' symmetric: coalesced
        }
        eql[KP * KP + lane] = yv;
    }
    if constexpr (NB <= 2) {
        if (!do_solve || cnt_real || lvl_count[l] == 0) return;   // level without samples: the reference never visits it (:147)
        wave_sync();
        double b = lane < KP ? yv : 0.0;
        double row[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) row[c] = lane < KP ? s_A[c * KP + lane] : 0.0;
#pragma unroll
        for (int c = 0; c < KP; ++c) row[c] += (c == lane && lane < K) ? lambda : 0.0;      // :174,187
        if (!gj_solve_regs<KP>(row, K, b, lane)) {                                          // :175,190; else the general route
            wave_sync();
            if (lane < K) s_A[lane * KP + lane] += lambda;
            b = lane < KP ? yv : 0.0;
            wave_sync();
            if (!lu_solve_lds(s_A, KP, K, b, lane)) { if (lane == 0) *fail = 1; return; }
        }
        if (lane < K) Arows[(size_t)l * KP + lane] = b;
    }
}

}  // namespace insider
