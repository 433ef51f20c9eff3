// insider_col_factored.hpp — the column-side masked Gram / XtY complement statistics without one rank-one update per
// held-out entry: categorical covariates only.
//
// For gene j with held-out samples H(j) and row factor rows r_i = sum_m a^m[level_m(i)] (src/optimize.cpp:216-222):
//   Gc_j = sum_{i in H(j)} r_i r_i' = M + M',   M = sum_o sum_{l in levels(o)} a^o_l p_{jl}',
//   p_{jl} = 1/2 n_{jl} a^o_l + sum_{i in l ∩ H(j)} sum_{m after o} a^m[level_m(i)],      n_{jl} = |l ∩ H(j)|,
// with the covariates taken in order of decreasing level count ("m after o").  (Expanding r r' over the covariates gives
// the squares a^m a^m' and the ordered cross terms a^o a^m'; M holds half of every square and one of each pair of cross
// terms.)  So the K x K work is one rank-one term a p' per (covariate, level) — 110 per gene at c3 against ~1000
// held-out entries — and every held-out entry costs one K-vector add of a row of a SMALL table (the covariates after
// the first have few levels; their rows live in LDS).  The x-dependent statistics need no pass at all:
//   qc_j = sum_{i in H(j)} x_ij r_i = sum_l A_l' (S - S^train)[j][l],    sum_{H(j)} x^2 = yy_all - yy_train.
// Output: the same record as k_list_stats with base = R'R (lower 16x16 blocks: XtX_j = R'R - Gc in the K x K part, qc in
// row KP-1 and the sum of squares in its corner), so the solve kernels do not change.  The level-grouped entry lists are those of the
// merged row update (insider_row_merged.hpp).
#pragma once

namespace insider {

constexpr int CF_MAXC = 8;        // covariates
constexpr int CF_CAP = 2048;      // uint16 look-up indices staged per wave (entries x later covariates)
constexpr int CP_MAXSTEPS = 8;    // pair-count form: table rows / 4 (k-steps of the count product)
constexpr int CP_MAXCELLS = 4096; // pair-count form: count bytes one pass of the builder histograms in LDS

struct ColFacArgs {
    int p, K, c;
    size_t plane;                         // entries per plane of slev
    const uint32_t *grp[CF_MAXC];         // position t (descending level count): [p][L + 1]
    const uint16_t *slev[CF_MAXC];        // its planes (stacked level of every other covariate, covariate order)
    int L[CF_MAXC + 1], off[CF_MAXC + 1]; // levels and stacked offset of the covariate at position t (position c: the
    int nlater[CF_MAXC + 1];              // continuous columns as one pseudo-covariate of m "levels", see zt)
    int later_plane[CF_MAXC][CF_MAXC];    // plane index inside slev[t] of each later covariate
    int tab_skip_lo, tab_skip_n;          // stacked levels [lo, lo + n) (position 0) are not in the LDS table
    int tab_rows;                         // SLcat - tab_skip_n
    const double *Astack;                 // SLcat x KP
    const double *Qheld;                  // p x KP: sum_l A_l' (S - S^train)[j][l]
    const double *RtR;                    // KP x KP: the record's K x K part is R'R - Gc = XtX_j (see k_list_stats)
    const double *yy_all, *yy_train;
    double *stat;                         // [p][STAT]
    // pair-count form (k_col_paircnt): per gene and position t the dense counts n_j(l, q) of held-out entries in level
    // l of the covariate at t whose LATER covariate has table row q, one byte per cell, stored in MFMA A-operand order
    // [l / 16][lane = (q % 4) * 16 + l % 16][q / 4] with 4 or 8 bytes per lane (one dword / two dwords per lane and
    // block of 16 levels); static per data set (mask and levels only)
    const uint8_t *cnt;
    int cnt_stride, cnt_off[CF_MAXC + 1], nsteps;   // bytes per gene, offset of position t, ceil(tab_rows / 4)
    // 1/2 n_j(l), the held-out entries of gene j in level l of the covariate at position t, as floats in the order the
    // kernel's lanes want them: [block of 16 levels][l % 4][(l % 16) / 4] (lane quarter g4 reads the four levels g4 + 4 s
    // of a block as one 16-byte word), zero beyond the last level; static per data set (k_half_counts)
    const float *hn;
    int hn_stride, hn_off[CF_MAXC + 1];         // floats per gene, offset of position t
    // continuous covariates (ctns_confounder, m <= 4 columns; optimize_continuous_v2, src/optimize.cpp:76-137) on the same
    // form: with r_i = t_i + A_c' z_i (t_i the categorical part) the complement sum_{i in H(j)} r_i r_i' gains the cross terms
    // sum_l a_l zeta_jl' A_c + transpose, zeta_jl = sum_{i in l ∩ H(j)} z_i, and A_c' ZZ_j A_c, ZZ_j = sum_{i in H(j)} z_i z_i'.
    // Both are more count products: the continuous columns are table rows that come AFTER every categorical covariate, with
    // REAL-valued counts zeta_jl (so p_jl += A_c' zeta_jl), and they are one more position (c) of m pseudo-levels whose
    // p_k = sum_k' (1/2 ZZ_j[k][k']) a^c_k'.  zt holds those counts per gene in MFMA A-operand order, one k-step of four table
    // rows: [position offset + block of 16 levels][lane = k * 16 + l % 16]; static per data set (k_zt_build).  null: none.
    const double *zt;
    int zt_stride, zt_off[CF_MAXC + 1];         // doubles per gene, offset of position t
    int m, SLcat;                               // continuous columns; their factor rows are Astack[SLcat .. SLcat + m)
    int pos_cov[CF_MAXC];                       // covariate (column of the level table) at position t
    // split solves (k_col_paircnt): the long genes' records are formed first, from their list, so that their solve can start
    // while the statistics of the others are still running; the launch over all genes then skips them
    const int *list;                      // gene ids of this launch (null: all genes 0 .. p-1)
    const int *list_count;                // ... of which the first *list_count are processed (null with list == null)
    const uint16_t *skip_bkt;             // all-gene launch: skip gene j when skip_bkt[j] <= *skip_last (null: none)
    const int *skip_last;
};

// Gc = M + M' for the lower blocks, XtX_j = R'R - Gc; qc and the sum of squares go into row KP - 1 of the record
// rtr: R'R (global, or a copy in LDS); qh / ss: this lane's Qheld entries (column 16 bj + c16) and the sum of squares,
// loaded by the caller (early, so that their latency hides behind the MFMAs)
// PREMASKED: rtr holds R'R inside the K x K part and zeros outside (the caller's LDS copy), so no index tests here
template <int NB, bool PREMASKED = false>
__device__ __forceinline__ void cf_store(d4 (&acc)[NB][NB], double *tr, const ColFacArgs &a, int j, int lane,
                                         const double *rtr, const double (&qh)[NB], double ss)
{
    constexpr int KP = Geo<NB>::KP;
    const int g4 = lane >> 4, c16 = lane & 15;
    double *out = a.stat + (size_t)j * Geo<NB>::STAT;
    int blk = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj, ++blk) {
            // transpose M(bj, bi) through LDS: register r of lane l holds element ((l >> 4) + 4 r, l & 15)
#pragma unroll
            for (int r = 0; r < 4; ++r) tr[(g4 + 4 * r) * 17 + c16] = acc[bj][bi][r];
            wave_sync();
            d4 res;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ra = 16 * bi + g4 + 4 * r, cb = 16 * bj + c16;
                res[r] = acc[bi][bj][r] + tr[c16 * 17 + g4 + 4 * r];
                if constexpr (PREMASKED) res[r] = rtr[ra * KP + cb] - res[r];   // outside K x K: -(0 + 0)
                else if (ra < a.K && cb < a.K) res[r] = rtr[ra * KP + cb] - res[r];
            }
            wave_sync();
            if (bi == NB - 1 && g4 == 3) {   // global row KP - 1 = local row 15 = register 3 of lanes 48..63
                const int col = 16 * bj + c16;
                res[3] = col < a.K ? qh[bj] : (col == KP - 1 ? ss : 0.0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) out[blk * 256 + (g4 + 4 * r) * 16 + c16] = res[r];
        }
}

template <int NB, int WPB>
__global__ void __launch_bounds__(WPB * 64) k_col_factored(ColFacArgs a)
{
    constexpr int KP = Geo<NB>::KP, NBLK = Geo<NB>::NBLK;
    extern __shared__ double s_cf[];   // [tab_rows + 1][KP] table | per wave: 16 x 17 transpose scratch | per wave: CF_CAP uint16
    double *tab = s_cf;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *tr = tab + (size_t)(a.tab_rows + 1) * KP + (size_t)w * 16 * 17;
    uint16_t *stg = reinterpret_cast<uint16_t *>(tab + (size_t)(a.tab_rows + 1) * KP + (size_t)WPB * 16 * 17) + (size_t)w * CF_CAP;
    for (int i = threadIdx.x; i < (a.tab_rows + 1) * KP; i += WPB * 64) {   // + one all-zero row
        const int r = i / KP, k = i % KP;
        const int q = r < a.tab_skip_lo ? r : r + a.tab_skip_n;
        tab[i] = r < a.tab_rows ? a.Astack[(size_t)q * KP + k] : 0.0;
    }
    __syncthreads();
    const int j = blockIdx.x * WPB + w;
    if (j >= a.p) return;
    const int g4 = lane >> 4, c16 = lane & 15;
    d4 acc[NB][NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj < NB; ++bj) acc[bi][bj] = d4{0.0, 0.0, 0.0, 0.0};
    for (int t = 0; t < a.c; ++t) {
        const uint32_t *g = a.grp[t] + (size_t)j * (a.L[t] + 1);
        const int Lo = a.L[t], nl = a.nlater[t];
        const uint32_t e0 = g[0], e1 = g[Lo];
        // stage the look-up indices of this ordering (entries x later covariates) when they fit; else read them from memory
        const bool staged = nl > 0 && (size_t)(e1 - e0) * nl <= (size_t)CF_CAP;
        if (staged) {
            for (int k = 0; k < nl; ++k) {
                const uint16_t *src = a.slev[t] + (size_t)a.later_plane[t][k] * a.plane;
                for (uint32_t x = e0 + lane; x < e1; x += WAVE) stg[(size_t)k * (e1 - e0) + (x - e0)] = src[x];
            }
        }
        wave_sync();
        // the group bounds and factor rows of the NEXT four levels are fetched while the current four are processed
        uint32_t bn, en;
        double avn[NB];
        auto fetch = [&](int l0n) {
            const int lgn = l0n + g4;
            const bool vn = lgn < Lo;
            const int lc = vn ? lgn : Lo - 1;
            const uint32_t gb = g[lc], ge = g[lc + 1];
            bn = vn ? gb : 0;
            en = vn ? ge : 0;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const double x = a.Astack[(size_t)(a.off[t] + lc) * KP + 16 * bb + c16];
                avn[bb] = vn ? x : 0.0;
            }
        };
        fetch(0);
        for (int l0 = 0; l0 < Lo; l0 += 4) {
            const uint32_t b = bn, e = en;
            double av[NB], pr[NB];
            const double hn = 0.5 * (double)(e - b);
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                av[bb] = avn[bb];
                pr[bb] = hn * av[bb];
            }
            fetch(l0 + 4);
            // sixteen entries of the group at a time: lane c16 fetches the look-up index of entry xb + c16, the indices go
            // round the quarter-wave by DPP row broadcast, and the 16 x NB table reads are independent (no serial LDS chain);
            // slots beyond the group read the all-zero row of the table
            for (int k = 0; k < nl; ++k) {
                for (uint32_t xb = b; xb < e; xb += 16) {
                    const uint32_t x = xb + c16;
                    int q = a.tab_rows;
                    if (x < e) {
                        q = staged ? (int)stg[(size_t)k * (e1 - e0) + (x - e0)]
                                   : (int)a.slev[t][(size_t)a.later_plane[t][k] * a.plane + x];
                        q = q < a.tab_skip_lo ? q : q - a.tab_skip_n;
                    }
                    const int qo = q * KP + c16;
#define CF_ADD(U)                                                                                   \
    {                                                                                               \
        const int qu = __builtin_amdgcn_update_dpp(0, qo, 0x150 + (U), 0xf, 0xf, true) - (U) + c16; \
        _Pragma("unroll") for (int bb = 0; bb < NB; ++bb) pr[bb] += tab[qu + 16 * bb];              \
    }
                    CF_ADD(0) CF_ADD(1) CF_ADD(2) CF_ADD(3) CF_ADD(4) CF_ADD(5) CF_ADD(6) CF_ADD(7)
                    CF_ADD(8) CF_ADD(9) CF_ADD(10) CF_ADD(11) CF_ADD(12) CF_ADD(13) CF_ADD(14) CF_ADD(15)
#undef CF_ADD
                }
            }
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = 0; bj < NB; ++bj)
                    acc[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[bi], pr[bj], acc[bi][bj], 0, 0, 0);
        }
        wave_sync();
    }
    double qh[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) qh[bb] = a.Qheld[(size_t)j * KP + 16 * bb + (lane & 15)];
    cf_store<NB>(acc, tr, a, j, lane, a.RtR, qh, a.yy_all[j] - a.yy_train[j]);
    (void)NBLK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pair-count form of the same statistics: p_jl = 1/2 n_jl a_l + sum_q n_j(l, q) tab_q, i.e. P = 1/2 diag(n) A + N_j Tab
// with the DENSE count matrix N_j (levels of the covariate x rows of the table of the later covariates).  N_j depends
// on the mask and the level ids only: it is built once per data set (k_pair_count_build), and the per-entry work of
// k_col_factored (one LDS row read + K-vector add per held-out entry) becomes ceil(L / 16) x ceil(rows / 4) x NB MFMAs.
// Worth it when the cells are not many more than the entries (c3: 100 x 10 cells for ~1000 entries per gene).
// One wave per gene.  16 levels at a time: P (16 x KP) = N_j-block x Tab by MFMA, + 1/2 n a in the accumulator layout
// (register r of lane (g4, c16) holds level g4 + 4 r — exactly the level that k-step r of the second product needs
// from that lane, so P feeds M += A' P with no data movement).
template <int NB>
struct PairBlk {          // operands of one block of 16 levels, raw as fetched (nothing here waits for the loads)
    uint32_t cw[2];       // this lane's count bytes (k-step s = byte s)
    float4 hn;            // 1/2 n of levels l0 + 4 s + g4, s = 0 .. 3 (0 beyond the last level)
    double av[4][NB];     // factor rows: k-step s holds level l0 + 4 s + g4, component 16 bb + c16
    double z;             // real-valued count of (level l0 + c16, continuous column g4) (ZC instantiation only)
};

// Round 3: every vector instruction of this kernel costs f64-MFMA issue time (section 4 of DESIGN.md), and a block of 16 levels
// used to spend ~75 of them around its 22 MFMAs (per-level address arithmetic with index clamps, group bounds -> counts ->
// doubles, zero weights for the levels beyond the last, LDS reads of the table operand).  Now:
//  * nothing is clamped or masked: rows beyond the covariate's last level are read as they come (the factor table has 16
//    rows of zero padding after its last row; the rows in between belong to the next covariate and are finite) and meet
//    EXACT ZEROS in P (their count bytes and their 1/2 n are zero), so they contribute nothing;
//  * 1/2 n comes from a static float table laid out for one 16-byte load per lane and block (ColFacArgs::hn);
//  * every address is a wave-uniform base (scalar arithmetic) plus a lane offset that never changes;
//  * the table operand of the count product lives in registers for the whole gene (2 x nsteps doubles per lane).
// ZC: the data set has continuous covariates (ColFacArgs::zt): one more k-step per block of 16 levels with real-valued
// counts, and one more position for the continuous columns themselves.  A separate instantiation: the categorical-only
// kernel keeps its register allocation.
template <int NB, int WPB, bool ZC = false>
__global__ void __launch_bounds__(WPB * 64) k_col_paircnt(ColFacArgs a)
{
    constexpr int KP = Geo<NB>::KP;
    extern __shared__ double s_cp[];   // per wave: 16 x 17 transpose scratch | R'R [KP][KP] | table rows [4 nsteps][KP], zero padded
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *tr = s_cp + (size_t)w * 16 * 17;
    double *rtr = s_cp + (size_t)WPB * 16 * 17;
    double *tabs = rtr + KP * KP;
    for (int i = threadIdx.x; i < KP * KP; i += WPB * 64) rtr[i] = (i / KP < a.K && i % KP < a.K) ? a.RtR[i] : 0.0;
    for (int i = threadIdx.x; i < 4 * a.nsteps * KP; i += WPB * 64) {
        const int r = i / KP, k = i % KP;
        const int q = r < a.tab_skip_lo ? r : r + a.tab_skip_n;
        tabs[i] = r < a.tab_rows ? a.Astack[(size_t)q * KP + k] : 0.0;
    }
    __syncthreads();
    int j = blockIdx.x * WPB + w;        // wave-uniform
    if (a.list) {
        if (j >= *a.list_count) return;
        j = a.list[j];
    } else {
        if (j >= a.p) return;
        if (a.skip_bkt && (int)a.skip_bkt[j] <= *a.skip_last) return;
    }
    const int g4 = lane >> 4, c16 = lane & 15;
    double qh[NB];                       // the epilogue's operands, requested now
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) qh[bb] = a.Qheld[(size_t)j * KP + 16 * bb + c16];
    const double ss = a.yy_all[j] - a.yy_train[j];
    const int bpl = a.nsteps <= 4 ? 4 : 8;   // count bytes per lane and block
    // B operand of the count product, k-step s, block bb: table row 4 s + g4, component 16 bb + c16
    double tb[CP_MAXSTEPS][NB];
#pragma unroll
    for (int s = 0; s < CP_MAXSTEPS; ++s)
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) tb[s][bb] = s < a.nsteps ? tabs[(4 * s + g4) * KP + 16 * bb + c16] : 0.0;
    // B operand of the real-count product: the factor row of continuous column g4 (rows beyond m: the table's zero padding)
    double tbz[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) tbz[bb] = ZC ? a.Astack[(size_t)(a.SLcat + g4) * KP + 16 * bb + c16] : 0.0;
    d4 acc[NB][NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj < NB; ++bj) acc[bi][bj] = d4{0.0, 0.0, 0.0, 0.0};
    const uint8_t *cj = a.cnt + (size_t)j * a.cnt_stride;
    const float *hj = a.hn + (size_t)j * a.hn_stride;
    const double *zj = ZC ? a.zt + (size_t)j * a.zt_stride : nullptr;
    const unsigned off_c = (unsigned)lane * (unsigned)bpl, off_h = (unsigned)g4 * 4u, off_a = (unsigned)(g4 * KP + c16);
    const int npos = a.c + (ZC ? 1 : 0);
    for (int t = 0; t < npos; ++t) {
        const int Lo = a.L[t];
        const bool cross = a.nlater[t] > 0;   // wave-uniform
        const uint8_t *ct = cj + a.cnt_off[t];
        const float *ht = hj + a.hn_off[t];
        const double *At = a.Astack + (size_t)a.off[t] * KP;
        const double *zt = ZC ? zj + a.zt_off[t] : nullptr;
        auto fetch = [&](int l0, PairBlk<NB> &b) {
            if constexpr (ZC) b.z = zt[(size_t)(l0 >> 4) * 64 + lane];
            b.cw[0] = b.cw[1] = 0;
            if (cross) {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(ct + (size_t)(l0 >> 4) * 64 * bpl + off_c);
                b.cw[0] = src[0];
                if (bpl == 8) b.cw[1] = src[1];
            }
            b.hn = *reinterpret_cast<const float4 *>(ht + l0 + off_h);
            const double *Ab = At + (size_t)l0 * KP;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) b.av[s][bb] = Ab[off_a + 4 * s * KP + 16 * bb];
        };
        // one block of 16 levels from its fetched operands
        auto compute = [&](int l0, PairBlk<NB> &cur) {
            d4 P[NB];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) P[bb] = d4{0.0, 0.0, 0.0, 0.0};
            if (cross) {
#pragma unroll
                for (int s = 0; s < CP_MAXSTEPS; ++s)
                    if (s < a.nsteps) {   // wave-uniform
                        const double cv = (double)((cur.cw[s >> 2] >> (8 * (s & 3))) & 0xffu);
#pragma unroll
                        for (int bb = 0; bb < NB; ++bb)
                            P[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv, tb[s][bb], P[bb], 0, 0, 0);
                    }
            }
            if constexpr (ZC) {
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) P[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.z, tbz[bb], P[bb], 0, 0, 0);
            }
            const double hn[4] = {(double)cur.hn.x, (double)cur.hn.y, (double)cur.hn.z, (double)cur.hn.w};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) P[bb][s] = fma(hn[s], cur.av[s][bb], P[bb][s]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (l0 + 4 * s < Lo) {   // wave-uniform
#pragma unroll
                    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                        for (int bj = 0; bj < NB; ++bj)
                            acc[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.av[s][bi], P[bj][s], acc[bi][bj], 0, 0, 0);
                }
        };
        // two operand sets: the loads of the next block of 16 levels are in flight during the MFMAs of the current one
        PairBlk<NB> b0, b1;
        fetch(0, b0);
        for (int l0 = 0; l0 < Lo; l0 += 32) {
            if (l0 + 16 < Lo) fetch(l0 + 16, b1);
            compute(l0, b0);
            if (l0 + 16 < Lo) {
                if (l0 + 32 < Lo) fetch(l0 + 32, b0);
                compute(l0 + 16, b1);
            }
        }
    }
    cf_store<NB, true>(acc, tr, a, j, lane, rtr, qh, ss);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_col_paircnt with the SECOND product (M += a_l p_l', 16 of its 22 matrix instructions per 16 levels at c3) on
// v_mfma_f64_4x4x4, NB <= 2.  The 16x16x4 form sustains ~0.6 of the f64 matrix peak chip-wide, the 4x4x4 form ~0.9
// (tools/ubench_mfma4.hip).  Operand map of the instruction (k_list_stats4): A[b][i][k] in lane 16 k + 4 b + i, B[b][k][j] in
// lane 16 k + 4 b + j, D[b][i][j] in lane 16 i + 4 b + j, four independent blocks b.  With lane = (g4, c16 = 4 b + i):
//  * P as the count product leaves it (register s of lane (g4, c16) = level 4 s + g4, column c16) IS a B operand: k = g4, block b
//    owns columns 4 b .. 4 b + 3 of the 16;
//  * a factor-row register in the natural layout (lane (g4, c16) = level 4 s + g4, component c16) IS an A operand whose block b
//    owns components 4 b .. 4 b + 3: the instruction gives the four DIAGONAL 4 x 4 tiles of a 16 x 16 block.  The other twelve come
//    from the same rows read at component (c16 + 4 x) mod 16, x = 1 .. 3: block b then owns row tile (b + x) mod 4 — sixteen
//    instructions cover the 64 tiles of a 32 x 32 product once (a Latin square), and the rotated operands cost no vector
//    instruction: the factor rows (identical for every gene) are staged once per block in LDS, four levels x 16 components
//    contiguous, and each lane reads its four rotations from there.  (Rotating P instead is 12 v_mov_dpp per four levels on
//    the port the f64 matrix instructions hold.)
// acc[bi][bj][x] of lane (g4, b, j) = M[16 bi + 4 ((b + x) & 3) + g4][16 bj + 4 b + j].
// LDS: per wave two 16 x 17 tiles | R'R | table rows | factor rows [position][level / 4][bi][level % 4][16]
template <int NB, int WPB, int MAXS, bool ZC = false>
__global__ void __launch_bounds__(WPB * 64) k_col_paircnt4(ColFacArgs a, int nitems, unsigned *__restrict__ ticket, unsigned ticket_base, int npart, int cap)
{
    static_assert(NB <= 2, "accumulators of the 4x4x4 form");
    constexpr int KP = Geo<NB>::KP;
    extern __shared__ double s_c4[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *tr = s_c4 + (size_t)w * 16 * 17;
    double *rtr = s_c4 + (size_t)WPB * 16 * 17;
    double *tabs = rtr + KP * KP;
    double *rows = tabs + 4 * a.nsteps * KP;
    const int npos = a.c + (ZC ? 1 : 0);
    for (int i = threadIdx.x; i < KP * KP; i += WPB * 64) rtr[i] = (i / KP < a.K && i % KP < a.K) ? a.RtR[i] : 0.0;
    for (int i = threadIdx.x; i < 4 * a.nsteps * KP; i += WPB * 64) {
        const int r = i / KP, k = i % KP;
        const int q = r < a.tab_skip_lo ? r : r + a.tab_skip_n;
        tabs[i] = r < a.tab_rows ? a.Astack[(size_t)q * KP + k] : 0.0;
    }
    {
        int pb4 = 0;   // quads of levels in front of position t
        for (int t = 0; t < npos; ++t) {
            const int nq = (a.L[t] + 3) >> 2;
            const double *At = a.Astack + (size_t)a.off[t] * KP;   // rows beyond the last level: the next covariate's or the zero padding
            for (int i = threadIdx.x; i < 4 * nq * KP; i += WPB * 64) {
                const int l = i / KP, k = i % KP;
                rows[((size_t)(pb4 + (l >> 2)) * NB + (k >> 4)) * 64 + (l & 3) * 16 + (k & 15)] = At[i];
            }
            pb4 += nq;
        }
        for (int i = threadIdx.x; i < 4 * KP; i += WPB * 64) rows[(size_t)pb4 * NB * 64 + i] = 0.0;   // the quad the look-ahead reads past the end
    }
    __syncthreads();
    const int g4 = lane >> 4, c16 = lane & 15;
    constexpr int bpl = MAXS <= 4 ? 4 : 8;   // count bytes per lane and block
    double tb[MAXS][NB];             // MAXS = 4 or 8 >= nsteps: the table operand costs 2 NB registers per k-step
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) tb[s][bb] = s < a.nsteps ? tabs[(4 * s + g4) * KP + 16 * bb + c16] : 0.0;
    double tbz[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) tbz[bb] = ZC ? a.Astack[(size_t)(a.SLcat + g4) * KP + 16 * bb + c16] : 0.0;
    const unsigned off_c = (unsigned)lane * (unsigned)bpl, off_h = (unsigned)g4 * 4u;
    // this lane's four rotations inside a quad of levels (doubles)
    const double *rl[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) rl[x] = rows + g4 * 16 + ((c16 + 4 * x) & 15);
    struct Blk {
        uint32_t cw[2];
        float4 hn;
        double z;
    };
    // A block keeps its staged rows for every gene its waves take (grid = the blocks that can be resident: the staging is paid
    // once).  Genes are dealt by TICKET: a static split would hand a block that starts late (another kernel holding its CU —
    // Qfull's product, a second fit on the same device) a full share to work off alone (measured: 0.28 -> 0.40 ms with a 20 us
    // product beside it).  One atomic per wave and gene, requested before the wave starts on its current gene.  Atomics on ONE
    // address retire at one per ~7 ns on this part — 50000 of them are 0.35 ms, more than the kernel — so there are npart
    // counters on lines of their own: block b draws from counter b % npart (late blocks are spread over all of them), whose
    // ticket t stands for gene t * npart + (b % npart), t < cap = ceil(nitems / npart).  The counters only ever grow: every
    // wave ends on exactly one ticket >= cap, the grid is a multiple of npart, so a launch takes cap + (waves per counter)
    // tickets from each and the host knows the next launch's base without a reset.
    const int part = blockIdx.x % npart;
    unsigned *tk = ticket + 32 * part;
    auto take = [&]() {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(tk, 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)v) - ticket_base;
    };
    for (unsigned cur = take(), nxt; cur < (unsigned)cap; cur = nxt) {
    nxt = take();
    if ((int)cur * npart + part >= nitems) continue;
    int j = (int)cur * npart + part;   // wave-uniform
    if (a.list) {
        if (j >= *a.list_count) continue;
        j = a.list[j];
    } else {
        if (a.skip_bkt && (int)a.skip_bkt[j] <= *a.skip_last) continue;
    }
    double qh[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) qh[bb] = a.Qheld[(size_t)j * KP + 16 * bb + c16];
    const double ss = a.yy_all[j] - a.yy_train[j];
    double acc[NB][NB][4];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj < NB; ++bj)
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[bi][bj][x] = 0.0;
    const uint8_t *cj = a.cnt + (size_t)j * a.cnt_stride;
    const float *hj = a.hn + (size_t)j * a.hn_stride;
    const double *zj = ZC ? a.zt + (size_t)j * a.zt_stride : nullptr;
    int pb4 = 0;
    for (int t = 0; t < npos; ++t) {
        const int Lo = a.L[t];
        const bool cross = a.nlater[t] > 0;   // wave-uniform
        const uint8_t *ct = cj + a.cnt_off[t];
        const float *ht = hj + a.hn_off[t];
        const double *zt = ZC ? zj + a.zt_off[t] : nullptr;
        auto fetch = [&](int l0, Blk &b) {
            if constexpr (ZC) b.z = zt[(size_t)(l0 >> 4) * 64 + lane];
            b.cw[0] = b.cw[1] = 0;
            if (cross) {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(ct + (size_t)(l0 >> 4) * 64 * bpl + off_c);
                b.cw[0] = src[0];
                if (bpl == 8) b.cw[1] = src[1];
            }
            b.hn = *reinterpret_cast<const float4 *>(ht + l0 + off_h);
        };
        // the four rotations of the factor rows of quad s (0 .. 4: 4 = the next block's first) of the block of 16 levels at rp:
        // constant offsets from four addresses that move once per block
        const double *rp[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) rp[x] = rl[x] + (size_t)pb4 * NB * 64;
        auto rows_of = [&](int s, double (&av)[4][NB]) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int bi = 0; bi < NB; ++bi) av[x][bi] = rp[x][(s * NB + bi) * 64];
        };
        auto step = [&](const double (&av)[4][NB], double hn, const d4 (&P)[NB], int s) {
            double pr[NB];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) pr[bb] = fma(hn, av[0][bb], P[bb][s]);
#pragma unroll
            for (int bi = 0; bi < NB; ++bi)
#pragma unroll
                for (int bj = 0; bj < NB; ++bj)
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        acc[bi][bj][x] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[x][bi], pr[bj], acc[bi][bj][x], 0, 0, 0);
        };
        double avA[4][NB], avB[4][NB];   // the rows of the current quad and of the next one (in flight during the current one's products)
        auto compute = [&](int l0, const Blk &cur) {
            d4 P[NB];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) P[bb] = d4{0.0, 0.0, 0.0, 0.0};
            if (cross) {
#pragma unroll
                for (int s = 0; s < MAXS; ++s)
                    if (s < a.nsteps) {   // wave-uniform
                        const double cv = (double)((cur.cw[s >> 2] >> (8 * (s & 3))) & 0xffu);
#pragma unroll
                        for (int bb = 0; bb < NB; ++bb)
                            P[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cv, tb[s][bb], P[bb], 0, 0, 0);
                    }
            }
            if constexpr (ZC) {
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) P[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur.z, tbz[bb], P[bb], 0, 0, 0);
            }
            rows_of(1, avB);
            __builtin_amdgcn_sched_barrier(0);   // the look-ahead reads stay in front of the products they hide behind
            step(avA, (double)cur.hn.x, P, 0);
            if (l0 + 4 < Lo) {   // wave-uniform
                rows_of(2, avA);
            __builtin_amdgcn_sched_barrier(0);   // the look-ahead reads stay in front of the products they hide behind
                step(avB, (double)cur.hn.y, P, 1);
                if (l0 + 8 < Lo) {
                    rows_of(3, avB);
            __builtin_amdgcn_sched_barrier(0);   // the look-ahead reads stay in front of the products they hide behind
                    step(avA, (double)cur.hn.z, P, 2);
                    if (l0 + 12 < Lo) {
                        rows_of(4, avA);
            __builtin_amdgcn_sched_barrier(0);   // the look-ahead reads stay in front of the products they hide behind
                        step(avB, (double)cur.hn.w, P, 3);
                    }
                }
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) rp[x] += 4 * NB * 64;
        };
        Blk b0, b1;
        fetch(0, b0);
        rows_of(0, avA);
        for (int l0 = 0; l0 < Lo; l0 += 32) {
            if (l0 + 16 < Lo) fetch(l0 + 16, b1);
            compute(l0, b0);
            if (l0 + 16 < Lo) {
                if (l0 + 32 < Lo) fetch(l0 + 32, b0);
                compute(l0 + 16, b1);
            }
        }
        pb4 += (Lo + 3) >> 2;
    }
    // epilogue: Gc = M + M' block by block through one 16 x 17 tile; the record as cf_store writes it
    {
        const int b = c16 >> 2, jj = c16 & 3;
        double *out = a.stat + (size_t)j * Geo<NB>::STAT;
        int blk = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj, ++blk) {
#pragma unroll
                for (int x = 0; x < 4; ++x) tr[(4 * ((b + x) & 3) + g4) * 17 + 4 * b + jj] = acc[bi][bj][x];
                wave_sync();
                d4 res;
#pragma unroll
                for (int r = 0; r < 4; ++r) res[r] = tr[(g4 + 4 * r) * 17 + c16];
                wave_sync();
#pragma unroll
                for (int x = 0; x < 4; ++x) tr[(4 * ((b + x) & 3) + g4) * 17 + 4 * b + jj] = acc[bj][bi][x];
                wave_sync();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ra = 16 * bi + g4 + 4 * r, cb = 16 * bj + c16;
                    res[r] = rtr[ra * KP + cb] - (res[r] + tr[c16 * 17 + g4 + 4 * r]);
                }
                wave_sync();
                if (bi == NB - 1 && g4 == 3) {
                    const int col = 16 * bj + c16;
                    res[3] = col < a.K ? qh[bj] : (col == KP - 1 ? ss : 0.0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) out[blk * 256 + (g4 + 4 * r) * 16 + c16] = res[r];
            }
    }
    }
}

// The real-valued count table of the continuous covariates (ColFacArgs::zt) and the held-out sums sum_{i in H(j)} x_ij z_ik,
// once per data set.  One wave per gene; every sum runs over the gene's held-out list in list order (fixed order:
// reproducible), each lane owning the levels l = lane, lane + 64, ... of a position.  lev: [c][n] zero-based levels; Zc: [m][n].
__global__ void __launch_bounds__(256) k_zt_build(ColFacArgs a, const uint32_t *__restrict__ col_ptr, const int *__restrict__ col_idx,
                                                  const double *__restrict__ col_val, const int *__restrict__ lev,
                                                  const double *__restrict__ Zc, int n, double *__restrict__ zt,
                                                  double *__restrict__ xz /*[p][xz_pitch], columns SLcat ..*/, int xz_pitch,
                                                  const double *__restrict__ S_all, double *__restrict__ S_train /*same pitch*/)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= a.p) return;
    const uint32_t eb = col_ptr[j], ee = col_ptr[j + 1];
    double *zo = zt + (size_t)j * a.zt_stride;
    for (int t = 0; t < a.c; ++t) {
        const int *lv = lev + (size_t)a.pos_cov[t] * n;
        for (int l = lane; l < a.L[t]; l += WAVE) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (uint32_t e = eb; e < ee; ++e) {
                const int i = col_idx[e];
                if (i < n && lv[i] == l) {      // (list padding carries an index beyond n)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < a.m) s[k] += Zc[(size_t)k * n + i];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) zo[a.zt_off[t] + (l >> 4) * 64 + k * 16 + (l & 15)] = s[k];
        }
    }
    // position c: 1/2 ZZ_j[k][k'] at lane k' * 16 + k;  the held-out sums of x z
    if (lane < 16) {
        const int k = lane & 3, k2 = lane >> 2;
        double zz = 0.0, sx = 0.0;
        for (uint32_t e = eb; e < ee; ++e) {
            const int i = col_idx[e];
            if (i < n && k < a.m && k2 < a.m) {
                const double zk = Zc[(size_t)k * n + i];
                zz += zk * Zc[(size_t)k2 * n + i];
                if (k2 == 0) sx += col_val[e] * zk;
            }
        }
        if (k < a.m && k2 < a.m) zo[a.zt_off[a.c] + k2 * 16 + k] = 0.5 * zz;
        if (k2 == 0 && k < a.m) {
            xz[(size_t)j * xz_pitch + a.SLcat + k] = sx;
            // the train-entry sums of x z (merged row update of a continuous column): all entries minus the held-out ones
            S_train[(size_t)j * xz_pitch + a.SLcat + k] = S_all[(size_t)j * xz_pitch + a.SLcat + k] - sx;
        }
    }
}

// weights of continuous column k as a one-level covariate of the merged row update: w_j = sum_{i in H(j)} z_ik^2 = 2 x the
// stored 1/2 ZZ_j[k][k]
__global__ void __launch_bounds__(256) k_cont_weights(const double *__restrict__ zt, int zt_stride, int zt_off_c, int k, int p,
                                                      double *__restrict__ w)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < p) w[j] = 2.0 * zt[(size_t)j * zt_stride + zt_off_c + k * 16 + k];
}

// 1/2 n_j(l) in the kernel's order (ColFacArgs::hn); once per data set.  One thread per (gene, padded level).
__global__ void __launch_bounds__(256) k_half_counts(ColFacArgs a, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)a.p * a.hn_stride) return;
    const int j = (int)(i / a.hn_stride), r = (int)(i % a.hn_stride);
    int t = 0;
    while (t + 1 < a.c && r >= a.hn_off[t + 1]) ++t;
    const int x = r - a.hn_off[t], blk = x >> 4, g4 = (x >> 2) & 3, sx = x & 3;
    const int l = 16 * blk + 4 * sx + g4;
    float v = 0.0f;
    if (l < a.L[t]) {
        const uint32_t *g = a.grp[t] + (size_t)j * (a.L[t] + 1);
        v = 0.5f * (float)(g[l + 1] - g[l]);
    }
    out[i] = v;
}

// The dense pair counts of every gene (once per data set): one wave per gene, LDS histogram per covariate position.
// overflow: set when a cell exceeds one byte (the pair-count form is then not used).
template <int WPB>
__global__ void __launch_bounds__(WPB * 64) k_pair_count_build(ColFacArgs a, uint8_t *cnt, int *overflow)
{
    __shared__ uint32_t hist[WPB][CP_MAXCELLS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * WPB + w;
    if (j >= a.p) return;
    uint32_t *hw = hist[w];
    const int bpl = a.nsteps <= 4 ? 4 : 8;
    const int pass_levels = CP_MAXCELLS / (64 * bpl) * 16;   // levels whose cells fit the histogram
    for (int t = 0; t < a.c; ++t) {
        const int Lo = a.L[t], nl = a.nlater[t];
        if (nl == 0) continue;
        const uint32_t *g = a.grp[t] + (size_t)j * (Lo + 1);
        for (int lb = 0; lb < Lo; lb += pass_levels) {
            const int le = lb + pass_levels < Lo ? lb + pass_levels : Lo;
            const int cells = ((le - lb + 15) >> 4) * 64 * bpl;
            for (int i = lane; i < cells; i += WAVE) hw[i] = 0;
            wave_sync();
            for (int l = lb; l < le; ++l) {
                const uint32_t b = g[l], e = g[l + 1];
                const int ll = l - lb;
                for (int k = 0; k < nl; ++k) {
                    const uint16_t *src = a.slev[t] + (size_t)a.later_plane[t][k] * a.plane;
                    for (uint32_t x = b + lane; x < e; x += WAVE) {
                        int q = (int)src[x];
                        q = q < a.tab_skip_lo ? q : q - a.tab_skip_n;
                        atomicAdd(&hw[(((ll >> 4) * 64) + (q & 3) * 16 + (ll & 15)) * bpl + (q >> 2)], 1u);
                    }
                }
            }
            wave_sync();
            uint8_t *out = cnt + (size_t)j * a.cnt_stride + a.cnt_off[t] + (size_t)(lb >> 4) * 64 * bpl;
            bool big = false;
            for (int i = lane; i < cells; i += WAVE) {
                const uint32_t v = hw[i];
                big = big || v > 255u;
                out[i] = (uint8_t)(v > 255u ? 255u : v);
            }
            if (big) *overflow = 1;
            wave_sync();
        }
    }
}

// u_j[l] = sum_{r in l ∩ H(j)} sum_{m != i} V_j[level_m(r)] of the merged row update (insider_row_merged.hpp, k_gene_u)
// from the same dense pair counts: for the covariate at position t,
//   u[l] = sum_q n^t_j(l, q) V_j[q]  (its own table: the covariates after it)
//        + sum_{t' < t} sum_{l'} n^{t'}_j(l', column of (t, l)) V_j[level l' of t']  (the tables of the covariates before it,
//                                                                                     read transposed).
// One wave per gene, ~20 instructions per block of 16 levels instead of one look-up per held-out entry and other covariate.
constexpr int GU_BATCH = 8;   // blocks of 16 levels whose partial sums share one trip through LDS (k_gene_u_cnt)
template <int WPB>
__global__ void __launch_bounds__(WPB * 64) k_gene_u_cnt(ColFacArgs a, int t, int LP, const double *__restrict__ V, int SLP,
                                                         int SL, double *__restrict__ U)
{
    extern __shared__ double s_uc[];   // per wave: V row [SL] | out [LP] | GU_BATCH x 64 partials
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * WPB + w;
    if (j >= a.p) return;
    double *v = s_uc + (size_t)w * (SL + LP + GU_BATCH * WAVE), *out = v + SL, *red = out + LP;
    const int Lt = a.L[t];
    // the gene's row of V, without the covariate's own columns (never read: u sums over the OTHER covariates)
    for (int q = lane; q < SL; q += WAVE) v[q] = (q < a.off[t] || q >= a.off[t] + Lt) ? V[(size_t)j * SLP + q] : 0.0;
    for (int l = lane; l < LP; l += WAVE) out[l] = 0.0;
    wave_sync();
    const int g4 = lane >> 4, c16 = lane & 15;
    const int bpl = a.nsteps <= 4 ? 4 : 8;
    const uint8_t *cj = a.cnt + (size_t)j * a.cnt_stride;
    if (a.nlater[t] > 0) {   // its own table: rows = its levels, columns = table rows of the later covariates
        double vq[CP_MAXSTEPS];
#pragma unroll
        for (int s = 0; s < CP_MAXSTEPS; ++s) {
            const int r = 4 * s + g4;
            const int q = r < a.tab_skip_lo ? r : r + a.tab_skip_n;
            vq[s] = (s < a.nsteps && r < a.tab_rows) ? v[q] : 0.0;
        }
        const uint8_t *ct = cj + a.cnt_off[t];
        // GU_BATCH blocks of 16 levels at a time (round 4): their count loads are issued together and their partial sums make ONE
        // trip through LDS — one block per trip left seven dependent global loads and fourteen wave syncs per gene at c3.  The
        // sum of a level's four partials keeps its order: bit-identical results.
        for (int l0 = 0; l0 < Lt; l0 += 16 * GU_BATCH) {
            uint32_t cw[GU_BATCH][2];
#pragma unroll
            for (int bq = 0; bq < GU_BATCH; ++bq) {
                cw[bq][0] = cw[bq][1] = 0u;
                if (l0 + 16 * bq < Lt) {   // wave-uniform
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(ct + ((size_t)((l0 >> 4) + bq) * 64 + lane) * bpl);
                    cw[bq][0] = src[0];
                    if (bpl == 8) cw[bq][1] = src[1];
                }
            }
#pragma unroll
            for (int bq = 0; bq < GU_BATCH; ++bq)
                if (l0 + 16 * bq < Lt) {
                    double part = 0.0;
#pragma unroll
                    for (int s = 0; s < CP_MAXSTEPS; ++s)
                        if (s < a.nsteps)   // wave-uniform: the table's k-steps only
                            part = fma((double)((cw[bq][s >> 2] >> (8 * (s & 3))) & 0xffu), vq[s], part);
                    red[64 * bq + lane] = part;
                }
            wave_sync();
            for (int x = lane; x < 16 * GU_BATCH && l0 + x < Lt; x += WAVE) {
                const double *r = red + 64 * (x >> 4) + (x & 15);
                out[l0 + x] = ((r[0] + r[16]) + r[32]) + r[48];
            }
            wave_sync();
        }
    }
    for (int tp = 0; tp < t; ++tp) {   // the tables of the covariates before it: columns of (t, l), transposed
        const int Lp = a.L[tp];
        const uint8_t *ct = cj + a.cnt_off[tp];
        double acc[CP_MAXSTEPS];
#pragma unroll
        for (int s = 0; s < CP_MAXSTEPS; ++s) acc[s] = 0.0;
        const uint32_t *src0 = reinterpret_cast<const uint32_t *>(ct + (size_t)lane * bpl);
        uint32_t nx[2] = {src0[0], bpl == 8 ? src0[1] : 0u};
        for (int l0 = 0; l0 < Lp; l0 += 16) {
            const uint32_t cw[2] = {nx[0], nx[1]};
            if (l0 + 16 < Lp) {   // the next block's counts fly during this block's FMAs
                const uint32_t *src = reinterpret_cast<const uint32_t *>(ct + ((size_t)((l0 >> 4) + 1) * 64 + lane) * bpl);
                nx[0] = src[0];
                if (bpl == 8) nx[1] = src[1];
            }
            const double vv = l0 + c16 < Lp ? v[a.off[tp] + l0 + c16] : 0.0;
#pragma unroll
            for (int s = 0; s < CP_MAXSTEPS; ++s)
                if (s < a.nsteps) acc[s] = fma((double)((cw[s >> 2] >> (8 * (s & 3))) & 0xffu), vv, acc[s]);
        }
        const int qs = a.off[t];                                              // stacked index of level 0 of (t)
        const int c_lo = qs < a.tab_skip_lo ? qs : qs - a.tab_skip_n;          // its table column
#pragma unroll
        for (int s = 0; s < CP_MAXSTEPS; ++s)
            if (s < a.nsteps) {   // wave-uniform
                const double tot = row16_sum(acc[s]);
                const int l = 4 * s + g4 - c_lo;
                if (c16 == 0 && l >= 0 && l < Lt) out[l] += tot;
            }
        wave_sync();
    }
    wave_sync();
    if (a.zt) {   // continuous covariates: + sum_k zeta_j[l][k] V_j[SLcat + k]  (s_r carries A_c' z_r; ColFacArgs::zt)
        const double *zr = a.zt + (size_t)j * a.zt_stride + a.zt_off[t];
        for (int l = lane; l < Lt; l += WAVE) {
            double sacc = out[l];
            for (int k = 0; k < a.m; ++k) sacc = fma(zr[(l >> 4) * 64 + k * 16 + (l & 15)], v[a.SLcat + k], sacc);
            out[l] = sacc;
        }
        wave_sync();
    }
    for (int l = lane; l < LP; l += WAVE) U[(size_t)j * LP + l] = l < Lt ? out[l] : 0.0;
}

// The same quantity for CONTINUOUS column k as a one-level covariate with real-valued membership weights z_rk
// (optimize_continuous_v2, src/optimize.cpp:76-137, on the merged form): u_j = c_j' sum_{r in H(j)} z_rk s_r with s_r the
// row factor without column k's own contribution = sum_l V_j[l] zeta_j[l][k] over every categorical level
// + sum_{k' != k} ZZ_j[k][k'] V_j[SLcat + k'].  One wave per gene, fixed summation order.  U: [p][2].
template <int WPB>
__global__ void __launch_bounds__(WPB * 64) k_gene_uc(ColFacArgs a, int k, const double *__restrict__ V, int SLP, double *__restrict__ U)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * WPB + w;
    if (j >= a.p) return;
    const double *zr = a.zt + (size_t)j * a.zt_stride, *vj = V + (size_t)j * SLP;
    double s = 0.0;
    for (int t = 0; t < a.c; ++t)
        for (int l = lane; l < a.L[t]; l += WAVE) s = fma(zr[a.zt_off[t] + (l >> 4) * 64 + k * 16 + (l & 15)], vj[a.off[t] + l], s);
    if (lane < a.m && lane != k) s = fma(2.0 * zr[a.zt_off[a.c] + lane * 16 + k], vj[a.SLcat + lane], s);
    s = group_sum<64>(s, lane);
    if (lane == 0) { U[(size_t)j * 2] = s; U[(size_t)j * 2 + 1] = 0.0; }
}

// Sheld = S - Strain (once per data set)
__global__ void __launch_bounds__(256) k_sub(const double *__restrict__ a, const double *__restrict__ b, size_t n,
                                             double *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = a[t] - b[t];
}

}  // namespace insider
