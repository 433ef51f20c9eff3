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
constexpr int CHUNK = 128;      // elements per streaming step (2 per lane); line pitches are multiples of it

constexpr int LEVEL_CHUNK = 4;   // member samples per wave of k_level_partial (row update, stage 1)
#define INSIDER_ORDER_ROW 448    // (a macro as well: the sweep assembly of insider_cd_reg.hpp spells it in its load offsets)
constexpr int ORDER_ROW = INSIDER_ORDER_ROW;   // bytes per sweep in the coordinate-order table (see k_order_table)

constexpr int CODE_TRAIN = 1;   // bit 0 of a mask code: entry is in the train set
constexpr int CODE_TEST = 2;    // bit 1: entry is in the test set (neither bit: NA)

template <int NB>
struct Geo {
    static constexpr int KP = 16 * NB;               // padded row length of factor rows (>= K + 1)
    static constexpr int NBLK = NB * (NB + 1) / 2;   // lower-triangular 16x16 blocks
    static constexpr int STAT = NBLK * 256;          // doubles per unit of block-stored statistics
    static constexpr int ROW_BYTES = KP * 8;         // bytes per padded factor row
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
// SYRK on gathered rows over a line's precomputed held-out list
// ---------------------------------------------------------------------------------------------
// f64 MFMAs hold the SIMD's vector issue port for their whole 64 cycles (measured: a second wave's integer VALU
// stream makes no progress while one wave issues v_mfma_f64_16x16x4_f64 back to back), so every VALU instruction
// of this kernel costs MFMA time.  The held-out lists (element index, value) of every line are therefore built
// ONCE per data set (the masks never change; the reference re-runs find() per gene per iteration,
// src/optimize.cpp:216-218) and the hot loop is 3 VALU + 2 buffer loads + 2 LDS reads per 3 MFMAs:
//  * four list entries form a group: lane l feeds entry (l >> 4) of the 4-deep panel, column (l & 15) of each
//    16-wide block; A and B operands of the MFMA are the same registers (A[i][k] = f~_k[16bi+i], B[k][j] = f~_k[16bj+j]);
//  * factor rows are fetched with raw buffer loads: 32-bit offset = index * row bytes + column, no 64-bit address
//    math, and list padding (index LIST_PAD) lands out of range, which the hardware returns as 0;
//  * the value x rides in the LAST slot (KP - 1) of the padded row: one select on a fixed lane;
//  * groups go in batches of SYRK_GB with two register sets: batch i+1's gathers are in flight during batch i's MFMAs.
constexpr int SYRK_GB = 4;
constexpr int SYRK_BATCH = 4 * SYRK_GB;        // list entries per batch
constexpr int LIST_ALIGN = 2 * SYRK_BATCH;     // lists are padded to a multiple of two batches (ping-pong drain)
constexpr int LIST_BLOCK = 64;                 // list entries staged through LDS at a time (LIST_BLOCK / 64 per lane)
constexpr int LIST_PAD = 0x7FFFFF;             // padding index: index * row bytes is beyond any factor buffer

typedef int v2i __attribute__((ext_vector_type(2)));

template <int NB>
__device__ __forceinline__ void syrk_load(const int *li, const double *lx, int batch, __amdgpu_buffer_rsrc_t rsrc,
                                          double (&a)[SYRK_GB][NB], int lane)
{
    constexpr int RB = Geo<NB>::KP * 8;
    const int sub = lane >> 4, c16 = lane & 15;
#pragma unroll
    for (int u = 0; u < SYRK_GB; ++u) {
        const int idx = li[SYRK_BATCH * batch + 4 * u + sub];
        const double xv = lx[SYRK_BATCH * batch + 4 * u + sub];
        const int off = idx * RB + c16 * 8;   // v_mad_u32_u24
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const v2i w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 128 * b, 0);
            a[u][b] = __hiloint2double(w.y, w.x);
        }
        a[u][NB - 1] = c16 == 15 ? xv : a[u][NB - 1];
    }
}

template <int NB>
__device__ __forceinline__ void syrk_mfma(const double (&a)[SYRK_GB][NB], d4 (&acc)[Geo<NB>::NBLK])
{
#pragma unroll
    for (int u = 0; u < SYRK_GB; ++u) {
        int blk = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj, ++blk)
                acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][bi], a[u][bj], acc[blk], 0, 0, 0);
    }
}

// nbatch >= 2, even, batches of SYRK_BATCH staged entries.  All loop control is scalar (readfirstlane) and the
// MFMAs are unconditional: a branch the compiler cannot prove uniform would make it mirror the accumulators in
// VGPRs (24 v_accvgpr_read per batch).
template <int NB>
__device__ __forceinline__ void drain_syrk(const int *li, const double *lx, int nbatch, __amdgpu_buffer_rsrc_t rsrc,
                                           d4 (&acc)[Geo<NB>::NBLK], int lane)
{
    nbatch = __builtin_amdgcn_readfirstlane(nbatch);
    double a0[SYRK_GB][NB], a1[SYRK_GB][NB];
    syrk_load<NB>(li, lx, 0, rsrc, a0, lane);
    for (int b = 0; b < nbatch; b += 2) {
        syrk_load<NB>(li, lx, b + 1, rsrc, a1, lane);
        syrk_mfma<NB>(a0, acc);
        // clamped look-ahead: re-loading the last batch is harmless and keeps the loads unconditional
        syrk_load<NB>(li, lx, b + 2 < nbatch ? b + 2 : nbatch - 1, rsrc, a0, lane);
        syrk_mfma<NB>(a1, acc);
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
// elastic-net coordinate descent, covariance form (strong_coordinate_descent,
// src/coordinate_descent.cpp:56-127)
// ---------------------------------------------------------------------------------------------
// At the workloads of BASELINE.json a subproblem needs thousands of sweeps (sub_tol = 1e-5 absolute on losses of
// ~1e5), so the K-step sequential sweep is THE hot loop of the whole path.  Layout: a wavefront is split into
// 64 / W groups of W lanes (W = 16, 32 or 64 >= K); each group owns one gene, lane l of the group owns
// coordinate l.  All groups walk the same coordinate order (include/insider_perm.h: the order does not depend on
// the gene; it is read from a table of max_sweeps rows precomputed per outer iteration), so the coordinate index
// of a step is wave-uniform and the per-step work is ~15 vector instructions for 64 / W genes:
//   reference residual form                  here
//   u = dot(residual, X_k) + b_k XtX_kk :94  h_k, with h = Xty - offdiag(XtX) b kept per lane (XtX_kk b_k folded in)
//   soft threshold / (XtX_kk + l2)  :99-104  every lane computes its own candidate; lane k's is broadcast
//   residual -= d X_k              :106-109  h -= d * offdiag(XtX)[:, k]   (row k of the LDS copy, zero diagonal)
//   |pre_loss - iter_loss| > tol        :114  exact loss change of the sweep from per-lane start/end values:
//                                             dL = -1/2 sum_l db_l (g_l + g'_l) + penalties(b') - penalties(b)
struct CdParams {
    double lambda, alpha, tol;
    double la, l2;          // lambda * alpha, lambda * (1 - alpha): formed on the host so that they arrive in SGPRs
    double two_la = 0.0, inv_two_la = 0.0;   // 2 la and 1 / (2 la): the register-resident sweeps work on a scaled state
                                             // (insider_cd_reg.hpp); la > 0 there
    int max_sweeps;
    const uint8_t *order;   // [min(max_sweeps, INSIDER_PERM_PERIOD) + 1][ORDER_ROW], see k_order_table: sweep s reads row
                            // s mod INSIDER_PERM_PERIOD (include/insider_perm.h); the extra row is a copy of row 0 (look-ahead)
    // multi-pass solves (register-resident kernel, cold outer iterations): a pass covers sweeps [start_sweep, sweep_limit);
    // genes still running at sweep_limit save their state (beta, h, 1/D or 0) and are re-packed, by predicted remaining
    // length, into the waves of the next pass, which continues them bit-identically.  sweep_limit == 0: run to the end.
    int start_sweep = 0, sweep_limit = 0;
};

// 1 / (XtX_kk + lambda (1 - alpha)) of src/coordinate_descent.cpp:99-104.  A zero denominator means the coordinate's
// regressor vanishes on the gene's training samples (a latent dimension that a pure lasso, alpha = 1, has switched off
// in every gene makes the next row update return an exactly zero factor column): then u = 0 <= lambda alpha and the
// reference returns 0 without dividing; 0 here gives the same update instead of 0 * inf.
__device__ __forceinline__ double cd_rcp(double d) { return d > 0.0 ? 1.0 / d : 0.0; }

template <int W>
__device__ __forceinline__ double group_bcast(double v, int k, int lane)
{
    if constexpr (W == 64) {
        return readlane_d(v, k);
    } else {
        // ds_bpermute: 1 address op + 2 LDS-crossbar ops for all groups at once.  (Two v_readlane pairs plus a
        // half-wave select have lower latency but cost 9 issue slots; the kernel is issue-bound at 2+ waves/SIMD.)
        const int addr = ((lane & ~(W - 1)) + k) << 2;   // v_lshl_add_u32: one op (no width mask as in __shfl)
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_ds_bpermute(addr, lo);
        hi = __builtin_amdgcn_ds_bpermute(addr, hi);
        return __hiloint2double(hi, lo);
    }
}

// DPP lane moves (32-bit; fp64 values move as two halves)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// sum over each 16-lane row, result in every lane of the row (quad swaps, half-mirror, mirror: 4 DPP steps)
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_mov_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov_d<0x141>(v);   // row_half_mirror
    v += dpp_mov_d<0x140>(v);   // row_mirror
    return v;
}

// (Not on the matrix unit: the contraction index of every f64 MFMA is lane / 16 — the DPP row number itself — and the four
// blocks of v_mfma_f64_4x4x4 are 4-lane groups, so no chain of them adds up the 16 lanes of a row; checked on the device,
// tools/mfma_rowsum.hip.)
template <int W>
__device__ __forceinline__ double group_sum(double v, int lane)
{
    v = row16_sum(v);
    if constexpr (W == 16) return v;
    if constexpr (W == 32) {
        const double a = readlane_d(v, 0) + readlane_d(v, 16), b = readlane_d(v, 32) + readlane_d(v, 48);
        return lane < 32 ? a : b;
    } else {
        return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
    }
}

template <int W>
__device__ __forceinline__ double group_max(double v)
{
#pragma unroll
    for (int o = W / 2; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// Goff: this group's LDS block, K rows of pitch W: Goff[k * W + l] = XtX[k][l] with a ZERO diagonal; Gll = XtX[l][l].
// q = Xty_l; beta: warm start in / solution out; g_out = (Xty - XtX beta)_l at the solution.
// valid: lane owns a real coordinate of a real gene.  s_ord: per-wave LDS scratch of 64 ints.
// Returns this group's sweep count, negated when the sweep cap (not convergence) ended the solve.
template <int W>
__device__ __forceinline__ int cd_sweeps(const double *Goff, int *s_ord, int K, double Gll, double q, double &beta,
                                         double &g_out, bool valid, const CdParams &P, int lane)
{
    const int l = lane & (W - 1);
    const uint64_t gmask = W == 64 ? ~0ull : (((1ull << (W & 63)) - 1ull) << (lane & ~(W - 1)));
    const uint64_t lt = lanemask_lt(lane);
    const double la = P.lambda * P.alpha, l2 = P.lambda * (1.0 - P.alpha);
    const double aq = valid ? fabs(q) : 0.0;
    const double mx = group_max<W>(aq);
    const double thr = P.alpha * (2.0 * P.lambda - mx);                               // :74
    bool active = valid && !(aq < thr);
    if (!active) beta = 0.0;                                                          // :78
    double h = valid ? q : 0.0;                                                       // :79 in covariance form
    for (int m = 0; m < K; ++m) h -= Goff[m * W + l] * group_bcast<W>(beta, m, lane);
    const double rinv = cd_rcp(Gll + l2);
    double inv = active ? rinv : 0.0;
    bool run = (__ballot(valid) & gmask) != 0;   // group holds a gene
    int sweep = 0, my_sweeps = 0;
    double bfinal = beta, gfinal = 0.0;
    uint32_t ordv = lane < K ? P.order[lane] : 0u;
    while (__any(run)) {
        const uint32_t ordn = (lane < K && sweep + 1 < P.max_sweeps)
                                  ? P.order[(size_t)((sweep + 1) & (int)(INSIDER_PERM_PERIOD - 1)) * ORDER_ROW + lane] : 0u;
        // this sweep's coordinate list: the table row without the coordinates screened out in every group (:83)
        uint64_t am = __ballot(active);
        if constexpr (W == 32) am = (am | (am >> 32)) & 0xffffffffull;
        if constexpr (W == 16) { am |= am >> 32; am |= am >> 16; am &= 0xffffull; }
        const bool keep = lane < K && ((am >> ordv) & 1ull);
        const uint64_t kb = __ballot(keep);
        const int nk = __popcll(kb);
        if (keep) s_ord[__popcll(kb & lt)] = (int)ordv;
        if (lane == 0) s_ord[nk] = 0;   // dummy entry read by the last step's look-ahead
        wave_sync();
        const int ordc = s_ord[lane];   // lanes > nk hold stale entries: never used
        wave_sync();
        const double beta0 = beta, g0 = h - beta * Gll;
        if (nk > 0) {
            int k = __builtin_amdgcn_readlane(ordc, 0);
            double gk = Goff[k * W + l];
#pragma unroll 2
            for (int t = 0; t < nk; ++t) {                                            // :91
                // next step's coordinate and Gram row are fetched ahead: they do not depend on this step
                const int kn = __builtin_amdgcn_readlane(ordc, t + 1);                // entry nk is a harmless dummy
                const double gn = Goff[kn * W + l];
                const double cand = copysign(fmax(fabs(h) - la, 0.0) * inv, h);       // :94-104
                const double d = group_bcast<W>(cand - beta, k, lane);
                h = fma(-d, gk, h);                                                   // :106-107
                beta = l == k ? cand : beta;                                          // :108
                k = kn;
                gk = gn;
            }
        }
        ++sweep;
        const double g1 = h - beta * Gll, db = beta - beta0;
        const double term = -0.5 * db * (g0 + g1) + 0.5 * l2 * (beta * beta - beta0 * beta0) +
                            la * (fabs(beta) - fabs(beta0));
        const double dloss = group_sum<W>(term, lane);                                // :112 as a difference
        if (run) {
            bool finish = sweep >= P.max_sweeps;
            if (!finish && !(fabs(dloss) > P.tol)) {                                  // :114
                const bool viol = valid && !active && fabs(g1) > P.alpha * P.lambda;  // :118-119 (grad = -g)
                if ((__ballot(viol) & gmask) != 0) {                                  // :123
                    if (viol) { active = true; inv = rinv; }
                } else {
                    finish = true;                                                    // :120-121
                }
            }
            if (finish) {   // park the group: zero increments from now on
                my_sweeps = (sweep >= P.max_sweeps && fabs(dloss) > P.tol) ? -sweep : sweep;   // negative: stopped by the cap
                bfinal = beta;
                gfinal = g1;
                beta = 0.0;
                inv = 0.0;
                active = false;
                run = false;
            }
        }
        ordv = ordn;
    }
    beta = bfinal;
    g_out = gfinal;
    return my_sweeps;
}

// bytes between the code blocks of consecutive coordinates in the register-resident sweep (insider_cd_reg.hpp).  A block is
// 56 bytes long: one 64-byte instruction-cache line per block (64, 96 and 128 ... 256-byte spacings measured the same in rounds
// 2 - 4: the cost of the computed jump is not the number of lines fetched)
#ifndef INSIDER_REG_BLOCK
#define INSIDER_REG_BLOCK 64
#endif
#define INSIDER_REG3_BLOCK 80   // ... in its three-slot form (32 < K <= 48): the blocks are exactly this long and packed, 49 of them stay below 4 KiB

// Order table, one row of ORDER_ROW bytes per sweep s < nsweeps: bytes [0, 64): the K coordinates in ascending key
// order (order_mode 0) or 0..K-1 (cyclic); bytes [64, 128): 32 uint16 = coordinate * pitch_bytes (row offsets for
// the row16 kernel; K > 32: 64 of them, bytes [64, 192)).  Bytes [128, 128 + 8 (1 + KMAX)), K <= 30: the successor list of the
// register-resident kernel (insider_cd_reg.hpp) as ABSOLUTE code addresses, 64 bits each: entry 0 = the block of the sweep's first
// coordinate, entry 1 + k = the block visited after coordinate k, the exit block (index exit_block = KMAX) after the last one
// and for k >= K; a block's address is code_base + INSIDER_REG_BLOCK * its index, code_base = where the kernel that will run
// the sweeps keeps its table of blocks (published by a probe launch, reg_code_base).  K = 31, 32: 32-bit block OFFSETS from byte
// 128 on (33 pairs do not fit beside the kernel's own scalars).  32 < K <= 48 on that kernel (wide_rows = 0): 32-bit offsets
// (INSIDER_REG3_BLOCK apart) from byte 124 on, 1 + K of them, and 30 row offsets.  One thread per (sweep, coordinate): rank by counting.
// pair_base != 0 (K <= 30): the sweep is cut greedily, from its first position on, into PAIRS of consecutive coordinates of one
// coordinate slot and single coordinates, and routed through the kernel's blocks of two steps (insider_cd_reg.hpp: pair (a, b) of
// slot 0 at pair_base + 128 (16 a + b), of slot 1 at pair_base + 128 (256 + W (a - 16) + (b - 16)), W = exit_block - 16); entry
// 1 + l is then the block after the block that ENDS with coordinate l.  Same steps in the same order: the iterates do not change.
__global__ void __launch_bounds__(256) k_order_table(uint64_t seed, uint32_t iter, int K, int nsweeps, int order_mode,
                                                     int pitch_bytes, int exit_block, int wide_rows, unsigned long long code_base,
                                                     unsigned long long pair_base, uint8_t *__restrict__ order)
{
    __shared__ uint8_t by_rank[4][64];
    __shared__ uint32_t keys[4][64];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = t >> 6, l = t & 63, w = threadIdx.x >> 6;
    const bool live = s <= nsweeps;             // row nsweeps: the look-ahead row, the order of sweep nsweeps mod PERIOD
    uint8_t *row = order + (size_t)(live ? s : 0) * ORDER_ROW;
    // every coordinate's key once (K hashes per sweep, not K^2), ranks by counting over the shared copy
    const uint32_t key = insider_perm_key(insider_perm_base(seed, iter, (uint32_t)s), (uint32_t)l);
    keys[w][l] = key;
    __syncthreads();
    int rank = l;
    if (live && l < K && order_mode == 0) {
        rank = 0;
        for (int m = 0; m < K; ++m) rank += keys[w][m] < key;
    }
    if (l < K) by_rank[w][rank] = (uint8_t)l;
    __syncthreads();
    if (!live) return;
    // wide rows (K > 32 on the row16 kernel with three or four slots): 64 row offsets in bytes [64, 192) and no successor list,
    // which shares those bytes; 32 < K <= 48 on the register-resident kernel: the list (1 + K <= 49 dwords from byte 124), 30 row offsets
    const bool wide = wide_rows != 0;
    if (K <= 30) {
        unsigned long long *pr = reinterpret_cast<unsigned long long *>(row + 128);
        const unsigned long long bb = INSIDER_REG_BLOCK, exit_addr = code_base + bb * (unsigned)exit_block;
        if (l >= K) {
            row[l] = 0;
            if (l < 32) reinterpret_cast<uint16_t *>(row + 64)[l] = 0;
            if (l < exit_block) pr[1 + l] = exit_addr;
            return;
        }
        row[rank] = (uint8_t)l;
        if (rank < 32) reinterpret_cast<uint16_t *>(row + 64)[rank] = (uint16_t)(l * pitch_bytes);
        if (pair_base) {
            const uint8_t *ord = by_rank[w];
            const int W = exit_block - 16;
            auto pairs = [&](int a, int b) { return (a >> 4) == (b >> 4); };
            auto block_at = [&](int pos) -> unsigned long long {
                if (pos >= K) return exit_addr;
                const int a = ord[pos];
                if (pos + 1 < K && pairs(a, ord[pos + 1])) {
                    const int b = ord[pos + 1];
                    return pair_base + 128ull * (unsigned)(a < 16 ? a * 16 + b : 256 + (a - 16) * W + (b - 16));
                }
                return code_base + bb * (unsigned)a;
            };
            int pos = 0, len = 1;
            for (;;) {      // every lane walks the cut up to its own coordinate
                len = (pos + 1 < K && pairs(ord[pos], ord[pos + 1])) ? 2 : 1;
                if (rank < pos + len) break;
                pos += len;
            }
            pr[1 + l] = block_at(pos + len);          // (read by a block only when l is its last coordinate)
            if (rank == 0) pr[0] = block_at(0);
            return;
        }
        pr[1 + l] = rank + 1 < K ? code_base + bb * by_rank[w][rank + 1] : exit_addr;
        if (rank == 0) pr[0] = code_base + bb * (unsigned)l;
        return;
    }
    uint32_t *blk = reinterpret_cast<uint32_t *>(row + 128);
    // K = 31, 32 (KMAX = 32: offsets, INSIDER_REG_BLOCK apart): dword 0 at byte 128 = first block, dword 1 + k = the block after k.
    // Narrow rows for K > 32 (three-slot register kernel, K <= 48) hold the list one dword earlier — first block at byte 124,
    // successor of coordinate k at 128 + 4 k — so that 1 + 48 dwords fit the row; the kernel loads from byte 124 there
    const int shift = K > 32 ? 1 : 0, nrow = shift ? 30 : 32;
    const uint32_t bb = K > 32 ? (uint32_t)INSIDER_REG3_BLOCK : (uint32_t)INSIDER_REG_BLOCK;   // bytes per code block
    if (l >= K) {
        row[l] = 0;
        if (l < nrow || wide) reinterpret_cast<uint16_t *>(row + 64)[l] = 0;
        if (l < 47 + shift && !wide) blk[1 + l - shift] = (uint32_t)exit_block * bb;
        return;
    }
    row[rank] = (uint8_t)l;
    if (rank < nrow || wide) reinterpret_cast<uint16_t *>(row + 64)[rank] = (uint16_t)(l * pitch_bytes);
    if (wide) return;
    if (l < 47 + shift) blk[1 + l - shift] = rank + 1 < K ? (uint32_t)by_rank[w][rank + 1] * bb : (uint32_t)exit_block * bb;
    if (rank == 0) blk[0 - shift] = (uint32_t)l * bb;
}

// ---------------------------------------------------------------------------------------------
// wave-level Cholesky on an LDS matrix: solve(XtX, Xty, likely_sympd)
// (src/optimize.cpp:175,190,226,240).  A: full symmetric, row pitch KP; destroyed (L in the lower triangle,
// L' mirrored in the upper).  b: lane l < K holds b_l in, x_l out.  Returns false if not positive definite.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool chol_solve_lds(double *A, int KP, int K, double &b, int lane)
{
    const bool valid = lane < K;
    const int lm = lane >> 3, ll = lane & 7;   // 8 x 8 tile of the trailing matrix per step
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
        // trailing update A[m][l] -= L[m][j] L[l][j] for j < m, l < K, all 64 lanes busy (row j now holds L[.][j])
        for (int m0 = j + 1; m0 < K; m0 += 8) {
            const int m = m0 + lm;
            const double lmj = m < K ? A[j * KP + m] : 0.0;
            for (int l0 = j + 1; l0 < K; l0 += 8) {
                const int l = l0 + ll;
                if (m < K && l < K) A[m * KP + l] -= lmj * A[j * KP + l];
            }
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
// General fallback of solve(XtX, Xty, likely_sympd) (src/optimize.cpp:175,190,226,240; src/fit_interaction.cpp:54):
// Armadillo tries the Cholesky route first and, when the matrix is not positive definite, solves the general system by
// LU with partial pivoting.  Here: Gauss-Jordan elimination with partial (row) pivoting on an LDS matrix, one wave,
// lane r owns row r and b_r.  A: K x K used part of a row-major matrix of pitch KP, destroyed.  b: lane l < K holds
// b_l in, x_l out.  Returns false when a pivot column is exactly zero (singular to working precision).
// Reached only when the positive-definite route reports a non-positive pivot: the lambda = 0 updates on a
// semidefinite level system, never on the lambda > 0 path of the BASELINE configurations.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lu_solve_lds(double *A, int KP, int K, double &b, int lane)
{
    const bool valid = lane < K;
    bool used = !valid;     // rows beyond K never pivot
    int mycol = -1;         // the column this lane's row became the pivot row of
    for (int j = 0; j < K; ++j) {
        double best = used ? -1.0 : fabs(A[lane * KP + j]);
        int bi = lane;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {           // arg max |a_rj| over the unused rows, lowest row on ties
            const double ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (!(best > 0.0)) return false;              // wave-uniform
        const int pr = bi;
        const double f = (lane == pr || !valid) ? 0.0 : A[lane * KP + j] / A[pr * KP + j];
        // the pivot row has zeros in the earlier pivot columns, and it is not written in this step (f = 0 for its lane)
        for (int c = j + 1; c < K; ++c) A[lane * KP + c] -= f * A[pr * KP + c];
        b -= f * __shfl(b, pr, 64);
        if (lane == pr) { used = true; mycol = j; }
        wave_sync();
    }
    const double xv = mycol >= 0 ? b / A[lane * KP + mycol] : 0.0;
    wave_sync();
    if (mycol >= 0) A[mycol] = xv;                    // the matrix is dead: its first K entries carry x to lane order
    wave_sync();
    b = valid ? A[lane] : 0.0;
    return true;
}

// ---------------------------------------------------------------------------------------------
// wave-level solve of a symmetric positive definite system in REGISTERS (K <= KP <= 32): lane l holds row l of A
// and b_l; Gauss-Jordan elimination without pivoting (on an SPD matrix the pivots are the d_j of L D L', all > 0,
// and the elimination is as stable as Cholesky).  The pivot row reaches the other lanes through v_readlane into
// SGPRs, which feed the fused multiply-adds directly; no LDS, no barriers, one reciprocal per column.
// solve(XtX, Xty, likely_sympd) (src/optimize.cpp:175,190).  Returns false on a non-positive pivot.
// ---------------------------------------------------------------------------------------------
template <int KP>
__device__ __forceinline__ bool gj_solve_regs(double (&row)[KP], int K, double &b, int lane)
{
    bool ok = true;
    double diag = 1.0;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        if (j < K) {                                     // wave-uniform
            const double piv = readlane_d(row[j], j);
            ok = ok && piv > 0.0;
            const double rp = 1.0 / piv;
            const double f = lane == j ? 0.0 : row[j] * rp;
            diag = lane == j ? piv : diag;
#pragma unroll
            for (int c = j + 1; c < KP; ++c) row[c] = fma(-f, readlane_d(row[c], j), row[c]);
            b = fma(-f, readlane_d(b, j), b);
        }
    }
    b /= diag;
    return ok;
}

// ---------------------------------------------------------------------------------------------
// Kernel: complement statistics of every line from its held-out list (both sides of the path)
// ---------------------------------------------------------------------------------------------
// unit u, segment s of nseg: batch pairs [s * per, (s+1) * per) of line u's list.  Output: Geo<NB>::STAT doubles at
// stat[(s * units + u) * STAT], block-major [blk][16][16] of the lower blocks of sum f~ f~', f~ = [f_e, 0.., x_e].
template <int NB, int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_list_stats(const uint32_t *__restrict__ ptr, const int *__restrict__ lidx, const double *__restrict__ lval, int units,
             int nseg, const double *__restrict__ F, int64_t f_rows, double *__restrict__ stat,
             const double *__restrict__ base /*KP x KP or null*/, int K)
{
    constexpr int NBLK = Geo<NB>::NBLK;
    __shared__ int s_li[WPB][2][LIST_BLOCK];
    __shared__ double s_lx[WPB][2][LIST_BLOCK];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * WPB + w;
    if (item >= (int64_t)units * nseg) return;
    const int u = (int)(item % units), sg = (int)(item / units);
    d4 acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) acc[b] = d4{0.0, 0.0, 0.0, 0.0};
    const uint32_t e_begin = ptr[u], e_end = ptr[u + 1];
    const int total = __builtin_amdgcn_readfirstlane((int)((e_end - e_begin) / LIST_ALIGN));   // batch PAIRS in the line
    const int per = (total + nseg - 1) / nseg;
    const int b0 = sg * per, b1 = b0 + per < total ? b0 + per : total;
    if (b0 < b1) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(F), 0, (int)(f_rows * Geo<NB>::KP * 8), 0x00020000);
        const uint32_t first = e_begin + (uint32_t)b0 * LIST_ALIGN, last = e_begin + (uint32_t)b1 * LIST_ALIGN;
        const int nblk = __builtin_amdgcn_readfirstlane((int)((last - first + LIST_BLOCK - 1) / LIST_BLOCK));
        const int nbt = __builtin_amdgcn_readfirstlane((int)((last - first) / SYRK_BATCH));   // even
        // two list blocks in flight in registers (EPL entries per lane each); entries beyond the segment read as padding
        constexpr int EPL = LIST_BLOCK / WAVE;
        int ia[EPL], ib[EPL];
        double xa[EPL], xb[EPL];
        auto ld = [&](int blk, int (&ii)[EPL], double (&xx)[EPL]) {
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                const uint32_t e = first + (uint32_t)blk * LIST_BLOCK + (uint32_t)(t * WAVE + lane);
                const uint32_t ec = e < last ? e : last - 1;   // clamped address, value masked: loads stay unconditional
                const int iv = lidx[ec];
                const double xv = lval[ec];
                ii[t] = e < last ? iv : LIST_PAD;
                xx[t] = e < last ? xv : 0.0;
            }
        };
        ld(0, ia, xa);
        ld(1, ib, xb);
        for (int blk = 0; blk < nblk; ++blk) {
            int *li = s_li[w][blk & 1];
            double *lx = s_lx[w][blk & 1];
#pragma unroll
            for (int t = 0; t < EPL; ++t) { li[t * WAVE + lane] = ia[t]; lx[t * WAVE + lane] = xa[t]; }
#pragma unroll
            for (int t = 0; t < EPL; ++t) { ia[t] = ib[t]; xa[t] = xb[t]; }
            ld(blk + 2, ib, xb);
            wave_sync();
            const int left = nbt - blk * (LIST_BLOCK / SYRK_BATCH);
            drain_syrk<NB>(li, lx, left < LIST_BLOCK / SYRK_BATCH ? left : LIST_BLOCK / SYRK_BATCH, rsrc, acc, lane);
        }
    }
    // With `base` (column side of the solver: base = R'R) the K x K part of the record is base - complement, i.e. the
    // gene's XtX itself (src/optimize.cpp:218-219): the solve kernels then read ONE operand stream per element.  Row
    // KP - 1 (the x slot) keeps the complement sums.
    double *out = stat + (size_t)item * Geo<NB>::STAT;
    const int sub = lane >> 4, c16 = lane & 15;
    int blk = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj, ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ra = 16 * bi + sub + 4 * r, cb = 16 * bj + c16;
                double v = acc[blk][r];
                if (base && ra < K && cb < K) v = base[ra * Geo<NB>::KP + cb] - v;
                out[blk * 256 + (sub + 4 * r) * 16 + c16] = v;
            }
}

// ---------------------------------------------------------------------------------------------
// The same statistics on v_mfma_f64_4x4x4 (four independent 4x4x4 products per instruction), 16 <= K <= 31
// ---------------------------------------------------------------------------------------------
// The 16x16x4 form computes full 16 x 16 blocks: for K + 1 = 26 coordinates 3 blocks = 768 outputs per four entries where
// the lower triangle of a 4 x 4 tiling has 28 tiles = 448.  Both forms run at the matrix unit's 32 flop per cycle and SIMD
// (measured, tools/ubench_mfma4.hip: 16 cycles per 4x4x4 instruction, 64 per 16x16x4), so the finer tiling is worth
// NT (NT + 1) / 2 x 16 cycles per 16 entries against NBLK x 4 x 64: 448 against 768 at K = 25, 576 against 768 at K = 30.
// Operand map of the instruction (measured): A[b][i][k] sits in lane 16 k + 4 b + i, B[b][k][j] in lane 16 k + 4 b + j,
// D[b][i][j] in lane 16 i + 4 b + j.  The four blocks b take FOUR DIFFERENT GROUPS of four entries of a 16-entry batch and
// all compute the same tile (ti, tj): lane (k, b, i) holds entry e = 4 b + k and fetches its coordinates 4 t + i, t < NT
// (NT loads of 8 bytes per lane and batch, as many bytes as the 16x16x4 form reads), so the A operand of tile (ti, tj) is
// the register a[ti] and the B operand a[tj] — no data movement between the loads and the matrix unit.  x rides in
// coordinate K (the rows of F are zero from K on).  At the end the four blocks' partial sums are added (two DPP rotations
// per tile) and the record is assembled in LDS in the layout k_list_stats writes (lower 16 x 16 blocks, x row = KP - 1).
// What bounds it: every load instruction touches 16 rows where k_list_stats' touch 4 whole 128-byte lines.  With 8-byte loads (32
// bytes of each row per instruction, 3.5 x the cache-line look-ups) the kernel is bound by the texture addresser as much as by the
// matrix unit: c5's statistics 0.69 -> 0.61 ms per launch, not the 0.42 the instruction count promises.  So the rows are read
// from a COPY of F whose coordinates are interleaved in pairs of tiles (k_tile_perm: position 8 s + 2 i + h holds coordinate
// 8 s + 4 h + i; 10 us per launch at c5): a lane's coordinates 4 t + i of the tiles t = 2 s, 2 s + 1 are then 16 adjacent bytes, one
// load instruction fetches two tiles (64 bytes of each row) and the look-ups halve.  (Fetching the rows as k_list_stats does and
// transposing them through an LDS tile per batch — 8 + 8 + NT more instructions and a wave sync per batch — measured 0.74.)
constexpr int tiles4(int NT) { return NT * (NT + 1) / 2; }
typedef int v4i __attribute__((ext_vector_type(4)));

// Fp[r][8 s + 2 i + h] = F[r][8 s + 4 h + i]   (s < KP / 8, i < 4, h < 2)
__global__ void __launch_bounds__(256) k_tile_perm(const double *__restrict__ F, int64_t rows, int KP, double *__restrict__ Fp)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= rows * KP) return;
    const int c = (int)(o % KP), s8 = c >> 3, i = (c >> 1) & 3, hh = c & 1;
    Fp[o] = F[o - c + 8 * s8 + 4 * hh + i];
}

template <int NB, int NT>
__device__ __forceinline__ void syrk4_load(const int *li, const double *lx, int batch, __amdgpu_buffer_rsrc_t rsrc,
                                           double (&a)[NT], int lane, int K)
{
    constexpr int RB = Geo<NB>::KP * 8;
    const int k = lane >> 4, b = (lane >> 2) & 3, i = lane & 3;
    const int e = SYRK_BATCH * batch + 4 * b + k;
    const int idx = li[e];
    const double xv = lx[e];
    const int off = idx * RB + i * 16;          // (rows in k_tile_perm's order: tiles 2 s and 2 s + 1 of coordinate lane i are adjacent)
#pragma unroll
    for (int s2 = 0; s2 < (NT + 1) / 2; ++s2) {
        const v4i w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 64 * s2, 0);
        a[2 * s2] = __hiloint2double(w.y, w.x);
        if (2 * s2 + 1 < NT) a[2 * s2 + 1] = __hiloint2double(w.w, w.z);
    }
    const int tx = K >> 2;                      // wave-uniform
    const bool mine = i == (K & 3);
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = (t == tx && mine) ? xv : a[t];
}

template <int NT>
__device__ __forceinline__ void syrk4_mfma(const double (&a)[NT], double (&acc)[tiles4(NT)])
{
    int q = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ti], a[tj], acc[q], 0, 0, 0);
}

template <int NB, int NT>
__device__ __forceinline__ void drain_syrk4(const int *li, const double *lx, int nbatch, __amdgpu_buffer_rsrc_t rsrc,
                                            double (&acc)[tiles4(NT)], int lane, int K)
{
    nbatch = __builtin_amdgcn_readfirstlane(nbatch);
    double a0[NT], a1[NT];
    syrk4_load<NB, NT>(li, lx, 0, rsrc, a0, lane, K);
    for (int b = 0; b < nbatch; b += 2) {
        syrk4_load<NB, NT>(li, lx, b + 1, rsrc, a1, lane, K);
        syrk4_mfma<NT>(a0, acc);
        syrk4_load<NB, NT>(li, lx, b + 2 < nbatch ? b + 2 : nbatch - 1, rsrc, a0, lane, K);   // clamped look-ahead, as in drain_syrk
        syrk4_mfma<NT>(a1, acc);
    }
}

// Same contract as k_list_stats (same lists, same record) except that F arrives in k_tile_perm's coordinate order;
// 4 (NT - 1) <= K < 4 NT so that the x coordinate K is the last tile's
template <int NB, int NT, int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_list_stats4(const uint32_t *__restrict__ ptr, const int *__restrict__ lidx, const double *__restrict__ lval, int units,
              int nseg, const double *__restrict__ F, int64_t f_rows, double *__restrict__ stat,
              const double *__restrict__ base /*KP x KP or null*/, int K)
{
    constexpr int KP = Geo<NB>::KP, STAT = Geo<NB>::STAT;
    static_assert(NT <= 4 * NB, "tiles beyond the padded row");
    __shared__ int s_li[WPB][2][LIST_BLOCK];
    __shared__ double s_lx[WPB][2][LIST_BLOCK];
    __shared__ double s_rec[WPB][STAT];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * WPB + w;
    if (item >= (int64_t)units * nseg) return;
    const int u = (int)(item % units), sg = (int)(item / units);
    double acc[tiles4(NT)];
#pragma unroll
    for (int q = 0; q < tiles4(NT); ++q) acc[q] = 0.0;
    const uint32_t e_begin = ptr[u], e_end = ptr[u + 1];
    const int total = __builtin_amdgcn_readfirstlane((int)((e_end - e_begin) / LIST_ALIGN));   // batch PAIRS in the line
    const int per = (total + nseg - 1) / nseg;
    const int b0 = sg * per, b1 = b0 + per < total ? b0 + per : total;
    if (b0 < b1) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(F), 0, (int)(f_rows * KP * 8), 0x00020000);
        const uint32_t first = e_begin + (uint32_t)b0 * LIST_ALIGN, last = e_begin + (uint32_t)b1 * LIST_ALIGN;
        const int nblk = __builtin_amdgcn_readfirstlane((int)((last - first + LIST_BLOCK - 1) / LIST_BLOCK));
        const int nbt = __builtin_amdgcn_readfirstlane((int)((last - first) / SYRK_BATCH));   // even
        constexpr int EPL = LIST_BLOCK / WAVE;
        int ia[EPL], ib[EPL];
        double xa[EPL], xb[EPL];
        auto ld = [&](int blk, int (&ii)[EPL], double (&xx)[EPL]) {
#pragma unroll
            for (int t = 0; t < EPL; ++t) {
                const uint32_t e = first + (uint32_t)blk * LIST_BLOCK + (uint32_t)(t * WAVE + lane);
                const uint32_t ec = e < last ? e : last - 1;   // clamped address, value masked: loads stay unconditional
                const int iv = lidx[ec];
                const double xv = lval[ec];
                ii[t] = e < last ? iv : LIST_PAD;
                xx[t] = e < last ? xv : 0.0;
            }
        };
        ld(0, ia, xa);
        ld(1, ib, xb);
        for (int blk = 0; blk < nblk; ++blk) {
            int *li = s_li[w][blk & 1];
            double *lx = s_lx[w][blk & 1];
#pragma unroll
            for (int t = 0; t < EPL; ++t) { li[t * WAVE + lane] = ia[t]; lx[t * WAVE + lane] = xa[t]; }
#pragma unroll
            for (int t = 0; t < EPL; ++t) { ia[t] = ib[t]; xa[t] = xb[t]; }
            ld(blk + 2, ib, xb);
            wave_sync();
            const int left = nbt - blk * (LIST_BLOCK / SYRK_BATCH);
            drain_syrk4<NB, NT>(li, lx, left < LIST_BLOCK / SYRK_BATCH ? left : LIST_BLOCK / SYRK_BATCH, rsrc, acc, lane, K);
        }
    }
    // ---- the record: zero it, add up the four blocks of every tile, scatter the tiles (both triangles of a diagonal 16 x 16
    //      block, as k_list_stats stores them), then write it out with the base applied -----------------------------------
    double *rec = s_rec[w];
    for (int o = lane; o < STAT; o += WAVE) rec[o] = 0.0;
    wave_sync();
    {
        const int i = lane >> 4, j = lane & 3;
        const bool writer = (lane & 12) == 0;      // block b = 0 of each row: lanes 16 i + j
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj, ++q) {
                double v = acc[q];
                v += dpp_mov_d<0x124>(v);           // row_ror:4
                v += dpp_mov_d<0x128>(v);           // row_ror:8: every lane of the row now holds the sum over b for its j
                int ra = 4 * ti + i, cb = 4 * tj + j;
                // coordinate K is x: its row / column live at KP - 1 in the record; coordinates beyond K are zero padding
                const bool keep = writer && ra <= K && cb <= K && cb <= ra;
                ra = ra == K ? KP - 1 : ra;
                cb = cb == K ? KP - 1 : cb;
                if (keep) {
                    const int bi = ra >> 4, bj = cb >> 4;
                    rec[(bi * (bi + 1) / 2 + bj) * 256 + (ra & 15) * 16 + (cb & 15)] = v;
                    if (bi == bj) rec[(bi * (bi + 1) / 2 + bj) * 256 + (cb & 15) * 16 + (ra & 15)] = v;
                }
            }
    }
    wave_sync();
    double *out = stat + (size_t)item * STAT;
    for (int o = lane; o < STAT; o += WAVE) {
        // block blk = (bi, bj) in the order bi-major; element (r16, c16)
        const int blk = o >> 8, r16 = (o >> 4) & 15, c16 = o & 15;
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
        const int bj = blk - bi * (bi + 1) / 2;
        const int ra = 16 * bi + r16, cb = 16 * bj + c16;
        double v = rec[o];
        if (base && ra < K && cb < K) v = base[ra * KP + cb] - v;
        out[o] = v;
    }
}

// ---- building the lists (once per data set) -----------------------------------------------------------------
// number of held-out entries of every line (one wave per line)
__global__ void __launch_bounds__(256) k_count_heldout(const uint8_t *__restrict__ codes, int64_t pitch, int len,
                                                       int lines, int *__restrict__ cnt)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= lines) return;
    int c = 0;
    for (int i = lane; i < len; i += WAVE) c += !(codes[(size_t)j * pitch + i] & CODE_TRAIN);
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) cnt[j] = c;
}

// (element index, value) of every held-out entry of a line, ascending, padded to a multiple of LIST_ALIGN
__global__ void __launch_bounds__(256) k_fill_lists(const double *__restrict__ vals, const uint8_t *__restrict__ codes,
                                                    int64_t pitch, int len, int lines, const uint32_t *__restrict__ ptr,
                                                    int *__restrict__ lidx, double *__restrict__ lval,
                                                    uint8_t *__restrict__ lflag /*optional: 1 = test entry*/)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= lines) return;
    uint32_t pos = ptr[j];
    const uint64_t lt = lanemask_lt(lane);
    for (int base = 0; base < len; base += WAVE) {
        const int i = base + lane;
        const int cd = i < len ? codes[(size_t)j * pitch + i] : CODE_TRAIN;
        const bool held = !(cd & CODE_TRAIN);
        const uint64_t b = __ballot(held);
        if (held) {
            const uint32_t o = pos + (uint32_t)__popcll(b & lt);
            lidx[o] = i;
            lval[o] = vals[(size_t)j * pitch + i];
            if (lflag) lflag[o] = (cd & CODE_TEST) ? 1 : 0;
        }
        pos += (uint32_t)__popcll(b);
    }
    for (uint32_t o = pos + lane; o < ptr[j + 1]; o += WAVE) { lidx[o] = LIST_PAD; lval[o] = 0.0; if (lflag) lflag[o] = 0; }
}

// ---------------------------------------------------------------------------------------------
// Kernel: column update (optimize_col, src/optimize.cpp:200-253) from the per-gene complement statistics
// ---------------------------------------------------------------------------------------------
enum ColMode { COL_EVAL = 0, COL_CD = 1 };

// index of element (a, b) of the lower-block-stored symmetric statistics (see k_line_stats)
__device__ __forceinline__ int stat_index(int a, int b)
{
    if (a < b) { const int t = a; a = b; b = t; }
    const int bi = a >> 4, bj = b >> 4;
    int i = a & 15, j = b & 15;
    if (bi == bj && i < j) { const int t = i; i = j; j = t; }   // diagonal blocks are stored full: either works
    return (bi * (bi + 1) / 2 + bj) * 256 + i * 16 + j;
}

struct ColArgs {
    const double *stat;      // [p][stat_len] complement statistics of every gene, or null (tuning == 0)
    int stat_len;
    int p, K, KP;
    const double *RtR;       // KP x KP
    const double *Qfull;     // p x KP: R' x_j over ALL samples (from the per-level sums)
    double *C;               // p x KP: warm start in, solution out
    const double *yy;        // p: sum of x^2 over train entries (masked) or over all entries
    int mode;                // ColMode
    int checkpoint;          // also produce per-gene loss statistics
    CdParams cd;
    double *sse_train, *b2, *b1;   // p each (checkpoint only)
    double *sse_test;        // p; written when test_from_stats
    int test_from_stats;     // no NA entry in the data: held-out == test, so the test residuals follow from the statistics
    int *sweeps;             // p
    unsigned long long *sweep_bins;   // 256 counters: total sweeps of the launch, spread to keep the atomics cheap
    const int *gene_perm;    // launch slot -> gene (genes sorted by their last sweep count, longest first), or null
    // multi-pass solves (CdParams::sweep_limit / start_sweep)
    double *hsave, *isave;       // p x KP each: h and 1/D-or-0 of the genes a limited pass left unfinished
    const int *pass_count;       // launch covers the slots below *pass_count only (null: all p): a resumed pass continues the first
                                 // *pass_count entries of its order; the long-gene launch of a split solve takes the first n_long
    const int *slot_begin;       // first slot of the launch = *slot_begin (null: 0): the majority launch of a split solve starts at n_long
    int resume;                  // the launch continues solves that a limited pass stopped (state in hsave / isave)
    uint32_t *pass_slot;         // limited pass, per gene it processed: CD_PASS_DONE, or (bucket << 24 | rank in the bucket) of a gene it
                                 // leaves unfinished (bucket = its estimated remaining sweeps on a log scale, see k_pass_scatter)
    int *bucket_cnt;             // limited pass: CD_BUCKETS counters (zeroed by the host)
    int *cap_hits;               // [0] counter of the genes a solve stopped at max_sweeps without convergence (the reference has no
                                 // cap, src/coordinate_descent.cpp:86-114), [1] the longest solve: insider_hip_get_info("cap_hits" /
                                 // "max_gene_sweeps"); may be null
    // launch order of the NEXT solve, first half (what k_sched_bucket does), done by the register-resident solve kernel itself
    // when a gene finishes: key update from its sweep count, bucket, rank inside the bucket.  null: not fused
    int *sched_key, *sched_cnt, *sched_rank;
    uint16_t *sched_bkt;
    int sched_reset;
    unsigned long long *code_base = nullptr;   // probe launch of the register-resident kernel (K <= 30): the addresses of its table of
                                               // code blocks [0] and of its blocks of two steps [1] are stored here and nothing
                                               // else is done (insider_cd_reg.hpp)
};

constexpr int CD_BUCKETS = 192;                  // 8 per octave of the estimate (1 .. 2^20 sweeps), longest first
constexpr uint32_t CD_PASS_DONE = 0xFFFFFFFFu;
__device__ __forceinline__ int cd_bucket(int est) { return CD_BUCKETS - 1 - min(CD_BUCKETS - 1, (int)(8.0f * __log2f((float)max(est, 1)))); }

// Between two passes of a multi-pass solve: the genes the pass left unfinished, grouped by bucket (longest estimate first;
// the order inside a bucket is the order of the atomics, which only decides which genes share a wave, never a result),
// become the first *count_out entries of perm_out.  n_in: genes the pass processed (*count_in when it was itself resumed).
__global__ void __launch_bounds__(256) k_pass_scatter(const uint32_t *__restrict__ pass_slot, const int *__restrict__ bucket_cnt,
                                                      const int *__restrict__ perm_in, const int *__restrict__ count_in, int p,
                                                      int *__restrict__ perm_out, int *__restrict__ count_out)
{
    __shared__ int off[CD_BUCKETS];
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < CD_BUCKETS; ++b) { off[b] = run; run += bucket_cnt[b]; }
        if (blockIdx.x == 0) *count_out = run;
    }
    __syncthreads();
    const int n_in = count_in ? *count_in : p;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n_in) return;
    const int j = perm_in ? perm_in[slot] : slot;
    const uint32_t v = pass_slot[j];
    if (v != CD_PASS_DONE) perm_out[off[v >> 24] + (int)(v & 0xFFFFFFu)] = j;
}

template <int W, int WPB>
__global__ void __launch_bounds__(WPB * 64) k_cd_cols(ColArgs a)
{
    constexpr int GPW = 64 / W;
    __shared__ double s_G[WPB][GPW][W * W];
    __shared__ int s_ord[WPB][80];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane / W, l = lane & (W - 1);
    const int slot = (blockIdx.x * WPB + w) * GPW + grp;
    const int j = slot < a.p ? (a.gene_perm ? a.gene_perm[slot] : slot) : a.p;
    const int K = a.K, KP = a.KP;
    const bool gene = j < a.p, valid = gene && l < K;
    double *Goff = s_G[w][grp];
    const double *st = (a.stat && gene) ? a.stat + (size_t)j * a.stat_len : nullptr;
    // XtX_j = R'R - complement (src/optimize.cpp:218-219), or the shared R'R (:234); zero diagonal in LDS
    double Gll = 1.0;
    for (int k = 0; k < K; ++k) {
        double v = 0.0;
        if (valid) {
            v = st ? st[stat_index(k, l)] : a.RtR[k * KP + l];   // the record holds XtX_j itself (see k_list_stats)
        }
        if (k == l) { Gll = valid ? v : 1.0; v = 0.0; }
        Goff[k * W + l] = v;
    }
    double q = 0.0, beta = 0.0;
    if (valid) {
        q = a.Qfull[(size_t)j * KP + l];                                                // :222,235 via level sums
        if (st) q -= st[stat_index(KP - 1, l)];                                         // minus the held-out part
        beta = a.C[(size_t)j * KP + l];
    }
    wave_sync();
    double g = 0.0;
    int sweeps = 0;
    if (a.mode == COL_CD) {                                                             // :228,246
        sweeps = cd_sweeps<W>(Goff, s_ord[w], K, Gll, q, beta, g, valid, a.cd, lane);
        const bool capped = sweeps < 0;
        sweeps = capped ? -sweeps : sweeps;
        if (valid) a.C[(size_t)j * KP + l] = beta;
        if (gene && l == 0) {
            if (a.cap_hits) {
                if (capped) atomicAdd(a.cap_hits, 1);
                if (sweeps > a.cap_hits[1]) atomicMax(a.cap_hits + 1, sweeps);
            }
            a.sweeps[j] = sweeps;
            if (a.sweep_bins) atomicAdd(&a.sweep_bins[(blockIdx.x * WPB + w) & 255], (unsigned long long)sweeps);
        }
    }
    if (!a.checkpoint) return;
    // ---- loss statistics with the (updated) column: sum_train (x - r'b)^2 = yy - 2 b'q + b'XtX b = yy - b'(q + g)
    g = valid ? q - Gll * beta : 0.0;   // fresh g = q - XtX b (the swept one carries thousands of sweeps' round-off)
    for (int m = 0; m < K; ++m) g -= Goff[m * W + l] * group_bcast<W>(beta, m, lane);
    const double bqg = group_sum<W>(valid ? beta * (q + g) : 0.0, lane);
    const double sb2 = group_sum<W>(valid ? beta * beta : 0.0, lane);
    const double sb1 = group_sum<W>(valid ? fabs(beta) : 0.0, lane);
    double te = 0.0;
    if (a.test_from_stats && st) {
        // sum_test (x - r'b)^2 = sum_held x^2 - 2 b'qc + b'Gc b, with the complement Gram Gc = R'R - XtX_j:
        //   b'Gc b = b'(R'R b) - b'(q - g)
        double rb = 0.0;
        for (int m = 0; m < K; ++m) rb += (valid ? a.RtR[l * KP + m] : 0.0) * group_bcast<W>(beta, m, lane);
        const double qc = valid ? st[stat_index(KP - 1, l)] : 0.0;
        const double part = group_sum<W>(valid ? beta * (rb - (q - g) - 2.0 * qc) : 0.0, lane);
        te = st[stat_index(KP - 1, KP - 1)] + part;
    }
    if (gene && l == 0) {
        a.sse_train[j] = a.yy[j] - bqg;
        a.b2[j] = sb2;
        a.b1[j] = sb1;
        if (a.test_from_stats) a.sse_test[j] = te;
    }
}

// alpha == 0: per-gene ridge solve (src/optimize.cpp:224-226,237-240); one wave per gene
struct RidgeArgs {
    const double *stat;
    int stat_len;
    int p, K, KP;
    const double *RtR, *Qfull;
    double *C;
    const double *yy;
    double lambda;
    int solve;        // 0: evaluate only
    int checkpoint;
    double *sse_train, *b2, *b1;
    double *sse_test;
    int test_from_stats;
    int *fail;
    // the register-resident kernel (insider_ridge_reg.hpp) has no general route: a gene whose system is not positive
    // definite is marked (mark[j] = 1, *retry = 1) and k_ridge_cols runs again for the marked genes only
    int *mark, *retry;
    int only_marked;
};

template <int WPB>
__global__ void __launch_bounds__(WPB * 64) k_ridge_cols(RidgeArgs a)
{
    __shared__ double s_A[WPB][64 * 64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * WPB + w;
    if (j >= a.p) return;
    if (a.only_marked && !(*a.retry && a.mark[j])) return;
    const int K = a.K, KP = a.KP;
    const bool valid = lane < K;
    double *A = s_A[w];
    const double *st = a.stat ? a.stat + (size_t)j * a.stat_len : nullptr;
    auto load = [&]() {
        for (int k = 0; k < K; ++k)
            if (valid) A[k * K + lane] = st ? st[stat_index(k, lane)] : a.RtR[k * KP + lane];
        wave_sync();
    };
    load();
    double q = 0.0, beta = 0.0;
    if (valid) {
        q = a.Qfull[(size_t)j * KP + lane] - (st ? st[stat_index(KP - 1, lane)] : 0.0);
        beta = a.C[(size_t)j * KP + lane];
    }
    if (a.solve) {
        if (valid) A[lane * K + lane] += a.lambda;
        wave_sync();
        double b = q;
        bool ok = chol_solve_lds(A, K, K, b, lane);                                     // :226,240 likely_sympd
        if (!ok) {                                                                      // general route (lu_solve_lds)
            wave_sync();
            load();
            if (valid) A[lane * K + lane] += a.lambda;
            wave_sync();
            b = q;
            ok = lu_solve_lds(A, K, K, b, lane);
        }
        if (!ok) { if (lane == 0) *a.fail = 1; }
        else beta = b;
        if (valid) a.C[(size_t)j * KP + lane] = beta;
        if (a.checkpoint) { wave_sync(); load(); }
    }
    if (!a.checkpoint) return;
    double g = valid ? q : 0.0;
    for (int m = 0; m < K; ++m) g -= (valid ? A[m * K + lane] : 0.0) * readlane_d(beta, m);
    const double bqg = wave_sum(valid ? beta * (q + g) : 0.0);
    const double sb2 = wave_sum(valid ? beta * beta : 0.0);
    const double sb1 = wave_sum(valid ? fabs(beta) : 0.0);
    double te = 0.0;
    if (a.test_from_stats && st) {   // as in k_cd_cols
        double rb = 0.0;
        for (int m = 0; m < K; ++m) rb += (valid ? a.RtR[lane * KP + m] : 0.0) * readlane_d(beta, m);
        const double qc = valid ? st[stat_index(KP - 1, lane)] : 0.0;
        te = st[stat_index(KP - 1, KP - 1)] + wave_sum(valid ? beta * (rb - (q - g) - 2.0 * qc) : 0.0);
    }
    if (lane == 0) {
        a.sse_train[j] = a.yy[j] - bqg;
        a.b2[j] = sb2;
        a.b1[j] = sb1;
        if (a.test_from_stats) a.sse_test[j] = te;
    }
}

// sum over the TEST entries of one gene of (x - r_i' beta)^2 (evaluate(), src/utils.cpp:67), from the gene's held-out
// list with per-entry flags (data with NA entries: held-out = test + NA).  One wave per gene, one entry per lane,
// beta broadcast from LDS.
template <int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_test_sse_list(const uint32_t *__restrict__ ptr, const int *__restrict__ lidx, const double *__restrict__ lval,
                const uint8_t *__restrict__ lflag, int lines, const double *__restrict__ F /*rows of pitch KP*/,
                const double *__restrict__ B /*[lines][KP]*/, int K, int KP, double *__restrict__ sse_test)
{
    __shared__ double s_b[WPB][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * WPB + w;
    if (j >= lines) return;
    s_b[w][lane] = lane < K ? B[(size_t)j * KP + lane] : 0.0;
    wave_sync();
    double te = 0.0;
    for (uint32_t e = ptr[j] + lane; e < ptr[j + 1]; e += WAVE) {
        if (lflag[e]) {
            const double *row = F + (size_t)lidx[e] * KP;
            double dot = 0.0;
            for (int k = 0; k < K; ++k) dot += row[k] * s_b[w][k];
            const double r = lval[e] - dot;
            te += r * r;
        }
    }
    te = wave_sum(te);
    if (lane == 0) sse_test[j] = te;
}

// ---------------------------------------------------------------------------------------------
// Kernel: batched stand-alone CD from dense (XtX, Xty) in global memory (insider_hip_strong_cd)
// ---------------------------------------------------------------------------------------------
template <int W, int WPB>
__global__ void __launch_bounds__(WPB * 64)
k_cd_batch(const double *__restrict__ XtX, const double *__restrict__ Xty, const double *__restrict__ wstart, int K,
           int64_t nprob, CdParams cd, double *__restrict__ beta_out, int *__restrict__ sweeps_out)
{
    constexpr int GPW = 64 / W;
    __shared__ double s_G[WPB][GPW][W * W];
    __shared__ int s_ord[WPB][80];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane / W, l = lane & (W - 1);
    const int64_t b = ((int64_t)blockIdx.x * WPB + w) * GPW + grp;
    const bool prob = b < nprob, valid = prob && l < K;
    double *Goff = s_G[w][grp];
    double Gll = 1.0;
    for (int k = 0; k < K; ++k) {
        double v = valid ? XtX[(size_t)b * K * K + (size_t)k * K + l] : 0.0;
        if (k == l) { Gll = valid ? v : 1.0; v = 0.0; }
        Goff[k * W + l] = v;
    }
    wave_sync();
    const double q = valid ? Xty[(size_t)b * K + l] : 0.0;
    double beta = valid ? wstart[(size_t)b * K + l] : 0.0, g;
    const int sw = cd_sweeps<W>(Goff, s_ord[w], K, Gll, q, beta, g, valid, cd, lane);
    if (valid) beta_out[(size_t)b * K + l] = beta;
    if (prob && l == 0 && sweeps_out) sweeps_out[b] = sw < 0 ? -sw : sw;
}

// ---------------------------------------------------------------------------------------------
// small dense kernels around the two streaming passes
// ---------------------------------------------------------------------------------------------

// R[r][:] = sum_i Astack[off_i + level_i(r)][:]   (src/optimize.cpp:365-369), rows of pitch KP
// (+ sum_j z_rj U_j for the continuous covariates, src/optimize.cpp:289,372: U = rows SLcat.. of Astack)
__global__ void k_build_R(const int *__restrict__ lev /*c x n, 0-based*/, const int *__restrict__ lvl_off, int c, int n,
                          const double *__restrict__ Astack, int KP, const double *__restrict__ Zc /*m x n*/, int m,
                          int SLcat, double *__restrict__ R)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * KP) return;
    const int r = (int)(t / KP), k = (int)(t % KP);
    double s = 0.0;
    for (int i = 0; i < c; ++i) s += Astack[(size_t)(lvl_off[i] + lev[(size_t)i * n + r]) * KP + k];
    for (int j = 0; j < m; ++j) s += Zc[(size_t)j * n + r] * Astack[(size_t)(SLcat + j) * KP + k];
    R[t] = s;
}

// S[gene][SLcat + j] = sum_r z_rj x_r,gene   (continuous covariates: "levels" with real-valued membership)
__global__ void __launch_bounds__(256) k_cont_sums(const double *__restrict__ vals, int64_t pitch, int p,
                                                   const double *__restrict__ Zc, int m, int n, int SLcat, int SLP,
                                                   double *__restrict__ S)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)p * m) return;
    const int g = (int)(t / m), j = (int)(t % m);
    double s = 0.0;
    for (int r = 0; r < n; ++r) s += Zc[(size_t)j * n + r] * vals[(size_t)g * pitch + r];
    S[(size_t)g * SLP + SLcat + j] = s;
}

// ---- gene scheduling for the sweep kernel: longest-first launch order, genes of similar length sharing a wave -----------
// Four genes share a wave until the slowest is done, so the launch order groups genes by their expected sweep count:
// key = the gene's sweep count (x 16) smoothed over the outer iterations once the counts have settled (reset = the first
// iterations, whose counts fall by an order of magnitude each; measured at c3: packing by the last count alone wastes
// 12-19 % (max of four > mean), by the smoothed count 8-13 %); for the first solve of a data set, with no counts yet, the
// gene's sum of squares (float bits; genes that carry signal need more sweeps: 16 % waste against 37 % in natural order).
// The order is a bucket sort on a log scale — 64 buckets per octave for the integer keys (1.1 % apart, far below the
// noise of the prediction), 16 per octave for the float keys — with one integer atomic per gene: the order inside a
// bucket is the order of the atomics, which decides which genes share a wave and never a result.  Two launches per
// outer iteration (this replaces a library radix / merge sort: ~7 launches).
constexpr int SCHED_BUCKETS = 2048;
__device__ __forceinline__ int sched_bucket(int key, int float_bits)
{
    int b = float_bits ? (key >> 19) - (63 << 4)                          // float bits: 2^-64 .. 2^64, 16 per octave
                       : (__float_as_int((float)key) >> 17) - (127 << 6);   // integer: 1 .. 2^31, 64 per octave
    b = b < 0 ? 0 : (b > SCHED_BUCKETS - 1 ? SCHED_BUCKETS - 1 : b);
    return SCHED_BUCKETS - 1 - b;                                          // longest first
}

// sweeps != null: key[j] = reset ? 16 sweeps[j] : (key[j] + 16 sweeps[j]) / 2 (integer keys); else key[] is given
// (float bits when float_bits).  bkt / rank: the gene's bucket and its rank inside it; cnt: SCHED_BUCKETS counters, zero on entry.
__global__ void __launch_bounds__(256) k_sched_bucket(const int *__restrict__ sweeps, int p, int reset, int float_bits,
                                                      int *__restrict__ key, int *__restrict__ cnt,
                                                      uint16_t *__restrict__ bkt, int *__restrict__ rank)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p) return;
    int k = key[j];
    if (sweeps) {
        const int s16 = sweeps[j] * 16;
        k = reset ? s16 : (k + s16) / 2;
        key[j] = k;
    }
    const int b = sched_bucket(k, float_bits);
    bkt[j] = (uint16_t)b;
    rank[j] = atomicAdd(&cnt[b], 1);
}

// perm[start of the gene's bucket + its rank] = gene; every block forms the bucket starts itself (exclusive scan of the
// SCHED_BUCKETS counters in LDS).  Block 0 also clears cnt_next, the counters of the next use (two sets alternate), and
// picks the LONG genes of a split solve: whole buckets from the longest down while they number at most cap_long — a set that
// does not depend on the order of the atomics.  long_out[0] = their number n_long (they are perm[0 .. n_long)), long_out[1] =
// the last long bucket (-1: none).
__global__ void __launch_bounds__(256) k_sched_scatter(const int *__restrict__ cnt, int *__restrict__ cnt_next,
                                                       const uint16_t *__restrict__ bkt, const int *__restrict__ rank, int p,
                                                       int *__restrict__ perm, int cap_long, int *__restrict__ long_out)
{
    constexpr int PER = SCHED_BUCKETS / 256;
    __shared__ int off[SCHED_BUCKETS];
    __shared__ int tot[256];
    const int t = threadIdx.x;
    int loc[PER], run = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { loc[q] = run; run += cnt[t * PER + q]; }
    tot[t] = run;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {   // inclusive scan of the 256 thread totals
        const int v = t >= o ? tot[t - o] : 0;
        __syncthreads();
        tot[t] += v;
        __syncthreads();
    }
    const int base = tot[t] - run;
#pragma unroll
    for (int q = 0; q < PER; ++q) off[t * PER + q] = base + loc[q];
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int q = t; q < SCHED_BUCKETS; q += 256) cnt_next[q] = 0;
        // inclusive counts are non-decreasing in the bucket index: the last bucket whose inclusive count fits the cap
        int best = -1;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int incl = base + loc[q] + cnt[t * PER + q];
            if (incl <= cap_long && incl > 0) best = t * PER + q;
        }
        __syncthreads();
        tot[t] = best;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (t < o) tot[t] = tot[t] > tot[t + o] ? tot[t] : tot[t + o];
            __syncthreads();
        }
        if (t == 0 && long_out) {
            const int b = tot[0];
            long_out[0] = b >= 0 ? off[b] + cnt[b] : 0;
            long_out[1] = b;
        }
    }
    const int j = blockIdx.x * 256 + t;
    if (j < p) perm[off[bkt[j]] + rank[j]] = j;
}

// First column solve of a data set, no sweep counts yet: the genes' sums of squares as float bits (positive floats order
// like their bit patterns).
__global__ void __launch_bounds__(256) k_yy_key(const double *__restrict__ yy, int p, int *__restrict__ key)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p) return;
    key[j] = __float_as_int(fmaxf((float)yy[j], 0.0f));
}

// out[o] = sum_b part[b][o]  (fixed order: bitwise reproducible).  Block = 16 outputs x 16 strided groups of partial
// blocks (short dependent chains), then the 16 group sums are added in group order.
__global__ void __launch_bounds__(256) k_sum_partials(const double *__restrict__ part, int nblk, int len,
                                                      double *__restrict__ out)
{
    __shared__ double red[16][17];
    const int ol = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + ol;
    double s = 0.0;
    if (o < len) {
#pragma unroll 8   // eight loads in flight, summed in the same order: a chain of single loads takes ~3 us each beside the row phase's kernels
        for (int b = g; b < nblk; b += 16) s += part[(size_t)b * len + o];
    }
    red[g][ol] = s;
    __syncthreads();
    if (g == 0 && o < len) {
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) t += red[m][ol];
        out[o] = t;
    }
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
    const double *R;         // n x KP: the CURRENT row factor rows (all covariates)
    const int *lev;          // c x n, 0-based
    const int *lvl_off;      // c + 1 prefix offsets into the stacked levels
    int cov;                 // updating categorical covariate, or -1 for a continuous column
    int own_row;             // continuous column: its row of Astack
    const double *weights;   // continuous column: z_r per sample (null: membership weight 1)
    const int *chunk_begin;  // per chunk: range into members
    const int *chunk_end;
    const int *members;      // sample ids sorted by level (this covariate)
    int nchunks;
    const double *Astack;    // SL x KP
    double *part;            // [nchunks][STAT + 2*KP + 2]: sum w^2 Hc | v = sum w (Hc s - bc) | sum w s | sum w^2
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
    double v = 0.0, ssum = 0.0, w2sum = 0.0;
    // the statistics of member mi + 1 are loaded while member mi is processed
    auto load_stat = [&](int r, d4 (&h)[NBLK]) {
#pragma unroll
        for (int b = 0; b < NBLK; ++b) h[b] = d4{0.0, 0.0, 0.0, 0.0};
        if (!a.masked) return;
        for (int sg = 0; sg < a.nseg; ++sg) {
            const double *src = a.stat + ((size_t)sg * a.n + r) * STAT;
#pragma unroll
            for (int b = 0; b < NBLK; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) h[b][q] += src[b * 256 + (sub + 4 * q) * 16 + c16];
        }
    };
    const int m_begin = a.chunk_begin[ch], m_end = a.chunk_end[ch];
    const int nm = m_end - m_begin;                          // <= LEVEL_CHUNK
    // the chunk's member ids, own rows and weights are fetched once (lane t < nm holds member t): the dependent
    // chain members -> level id -> factor rows is paid per chunk, not per member
    int r_l = 0, own_l = 0;
    double w_l = 1.0;
    if (lane < nm) {
        r_l = a.members[m_begin + lane];
        own_l = a.cov >= 0 ? a.lvl_off[a.cov] + a.lev[(size_t)a.cov * a.n + r_l] : a.own_row;
        w_l = a.weights ? a.weights[r_l] : 1.0;
    }
    // s_r: everything but this covariate's own contribution (the Gauss-Seidel residual of :338,344 is x_r - s_r'C)
    double s_all[LEVEL_CHUNK], w_all[LEVEL_CHUNK];
#pragma unroll
    for (int t = 0; t < LEVEL_CHUNK; ++t) {
        const int r = __builtin_amdgcn_readlane(r_l, t), own = __builtin_amdgcn_readlane(own_l, t);
        w_all[t] = readlane_d(w_l, t);
        s_all[t] = (valid && t < nm) ? a.R[(size_t)r * KP + lane] - w_all[t] * a.Astack[(size_t)own * KP + lane] : 0.0;
    }
    d4 hn[NBLK];
    load_stat(__builtin_amdgcn_readlane(r_l, 0), hn);
#pragma unroll
    for (int t = 0; t < LEVEL_CHUNK; ++t) {
        if (t < nm) {                                        // wave-uniform
            const double w = w_all[t], s = s_all[t];
            ssum += w * s;
            w2sum += w * w;
            if (a.masked) {
                d4 h[NBLK];
#pragma unroll
                for (int b = 0; b < NBLK; ++b) h[b] = hn[b];
                load_stat(__builtin_amdgcn_readlane(r_l, t + 1 < nm ? t + 1 : t), hn);
                wave_sync();
                acc_to_lds<NB>(h, s_H, lane);
                if (lane < KP) s_s[lane] = s;
                wave_sync();
                if (valid) {
                    double y = -s_H[(KP - 1) * KP + lane];     // - bc_r (the x slot is the last one of the padded row)
                    for (int b = 0; b < K; ++b) y += s_H[b * KP + lane] * s_s[b];   // H symmetric: conflict-free column walk
                    v += w * y;
                }
                const double w2 = w * w;
#pragma unroll
                for (int b = 0; b < NBLK; ++b) hsum[b] += w2 * h[b];
            }
        }
    }
    double *out = a.part + (size_t)ch * (STAT + 2 * KP + 2);
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[b * 256 + (sub + 4 * q) * 16 + c16] = hsum[b][q];
    if (lane < KP) { out[STAT + lane] = valid ? v : 0.0; out[STAT + KP + lane] = valid ? ssum : 0.0; }
    if (lane == 0) { out[STAT + 2 * KP] = w2sum; out[STAT + 2 * KP + 1] = 0.0; }
}

// Stage 2a: per level, the sum of its chunk partials (vectors of `len` doubles) in a fixed order: block = 16 outputs x
// 16 strided groups of chunks, then the 16 group sums in group order.  grid = (ceil(len / 16), L).
__global__ void __launch_bounds__(256) k_level_sum(const double *__restrict__ part, const int *__restrict__ lvl_chunk_ptr,
                                                   int len, double *__restrict__ out /*[L][out_stride]*/, int out_stride)
{
    __shared__ double red[16][17];
    const int ol = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + ol, l = blockIdx.y;
    const int c0 = lvl_chunk_ptr[l], c1 = lvl_chunk_ptr[l + 1];
    double s = 0.0;
    if (o < len) {
#pragma unroll 8   // (loads in flight together, same summation order: see k_sum_partials)
        for (int ch = c0 + g; ch < c1; ch += 16) s += part[(size_t)ch * len + o];
    }
    red[g][ol] = s;
    __syncthreads();
    if (g == 0 && o < len) {
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) t += red[m][ol];
        out[(size_t)l * out_stride + o] = t;
    }
}

// Stage 2b (one wave per level): from the level's summed partials form this rank's share of
// the normal equations: eq[l] = {XtX (KP x KP, full symmetric, no ridge term), Xty (KP)}.
struct LevelReduceArgs {
    const double *part;         // [L][STAT + 2*KP + 2]: the level sums of k_level_sum
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
    double v = 0.0, ssum = 0.0;
    const double *src = a.part + (size_t)l * (STAT + 2 * KP + 2);
    const double cnt = src[STAT + 2 * KP];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) h[b][q] = src[b * 256 + (sub + 4 * q) * 16 + c16];
    if (lane < KP) { v = src[STAT + lane]; ssum = src[STAT + KP + lane]; }
    acc_to_lds<NB>(h, s_H, lane);
    if (lane < KP) s_s[lane] = ssum;
    wave_sync();
    double *eq = a.eq + (size_t)l * (KP * KP + KP);
    for (int i = lane; i < KP * KP; i += WAVE) {
        const int x = i / KP, y = i % KP;
        eq[i] = (x < K && y < K) ? cnt * a.CCt[i] - s_H[i] : 0.0;
    }
    if (lane < KP) {
        double y = 0.0;
        if (valid) {
            y = a.SC[(size_t)(a.sc_off + l) * KP + lane] + v;
            for (int b = 0; b < K; ++b) y -= a.CCt[b * KP + lane] * s_s[b];      // CC' symmetric: coalesced
        }
        eq[KP * KP + lane] = y;
    }
}

// Masked update of one continuous column (optimize_continuous_v2, src/optimize.cpp:102-126) on its reduced
// normal equations eq = {H (KP x KP, no ridge term), b (KP)}: cyclic scalar coordinate descent
// u_i = (b_i - sum_{l != i} H_il u_l) / (H_ii + lambda) until sum |u - u_previous_pass| < 0.1.  One wave.
template <int NB>
__global__ void __launch_bounds__(64) k_cont_cd(const double *__restrict__ eq, int K, double lambda,
                                                double *__restrict__ urow /*KP*/)
{
    constexpr int KP = Geo<NB>::KP;
    __shared__ double s_A[KP * KP];
    const int lane = threadIdx.x;
    for (int i = lane; i < KP * KP; i += WAVE) s_A[i] = eq[i];
    wave_sync();
    const bool valid = lane < K;
    const double b = valid ? eq[KP * KP + lane] : 0.0;
    double u = valid ? urow[lane] : 0.0;
    for (int pass = 0; pass < 100000; ++pass) {                                         // :102 while(1)
        const double pre = u;
        for (int i = 0; i < K; ++i) {                                                   // :104
            const double dot = wave_sum(valid && lane != i ? s_A[i * KP + lane] * u : 0.0);
            const double bi = readlane_d(b, i), hii = s_A[i * KP + i];
            const double ui = (bi - dot) / (hii + lambda);                              // :117
            if (lane == i) u = ui;
        }
        if (wave_sum(valid ? fabs(pre - u) : 0.0) < 1e-1) break;                        // :122
    }
    if (valid) urow[lane] = u;
}

// Stage 3 (one wave per level, after the cross-rank sum): add the ridge term and solve; write A_i[l].
// solve(..., likely_sympd): the positive-definite route first (register Gauss-Jordan / LDS Cholesky); on a non-positive
// pivot the general route (lu_solve_lds) on a fresh copy of the system; `fail` only when that is singular too.
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
    bool ok;
    double b = lane < KP ? src[KP * KP + lane] : 0.0;
    if constexpr (NB <= 2) {
        // row `lane` of XtX in registers (the matrix is symmetric: column walks are coalesced)
        double row[KP];
#pragma unroll
        for (int c = 0; c < KP; ++c) row[c] = lane < KP ? src[c * KP + lane] : 0.0;
#pragma unroll
        for (int c = 0; c < KP; ++c) row[c] += (c == lane && lane < K) ? lambda : 0.0;      // :174,187
        ok = gj_solve_regs<KP>(row, K, b, lane);                                            // :175,190
    } else {
        for (int i = lane; i < KP * KP; i += WAVE) s_A[i] = src[i];
        wave_sync();
        if (lane < K) s_A[lane * KP + lane] += lambda;                                      // :174,187
        wave_sync();
        ok = chol_solve_lds(s_A, KP, K, b, lane);                                           // :175,190
    }
    if (!ok) {                                                                              // general route
        wave_sync();
        for (int i = lane; i < KP * KP; i += WAVE) s_A[i] = src[i];
        wave_sync();
        if (lane < K) s_A[lane * KP + lane] += lambda;
        b = lane < KP ? src[KP * KP + lane] : 0.0;
        wave_sync();
        if (!lu_solve_lds(s_A, KP, K, b, lane)) { if (lane == 0) *fail = 1; return; }
    }
    if (lane < K) Arows[(size_t)l * KP + lane] = b;
}

// X'X (K x K, column-major) and X'y of an m x K column-major design matrix and outcome: what the reference's callers
// hand to strong_coordinate_descent next to (X, y) (src/optimize.cpp:219-222,228; :234-235,246).  One wave per output
// entry (a, b), b == K selects y; fixed-order reduction.
__global__ void __launch_bounds__(64) k_xtx_xty(const double *__restrict__ X, const double *__restrict__ y, int64_t m, int K,
                                                double *__restrict__ XtX, double *__restrict__ Xty)
{
    const int a = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const double *xa = X + (size_t)a * m, *xb = b < K ? X + (size_t)b * m : y;
    double acc = 0.0;
    for (int64_t i = lane; i < m; i += WAVE) acc = fma(xa[i], xb[i], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
        if (b < K) XtX[a + (size_t)b * K] = acc;
        else Xty[a] = acc;
    }
}

// solve(A, b, likely_sympd) on its own, batched: one wave per K x K system (column-major = row-major of A', and the
// positive-definite route needs A symmetric; the general route solves A' x = b for the transposed read, so the matrix
// is loaded transposed to keep A x = b for non-symmetric input).  route[s]: 0 = Cholesky, 1 = general, -1 = singular.
__global__ void __launch_bounds__(64) k_solve_batch(const double *__restrict__ A, const double *__restrict__ bvec, int K,
                                                    int64_t nsys, double *__restrict__ x, int *__restrict__ route)
{
    __shared__ double s_A[64 * 64];
    const int64_t sidx = blockIdx.x;
    const int lane = threadIdx.x;
    if (sidx >= nsys) return;
    const double *src = A + (size_t)sidx * K * K;
    auto load = [&]() {
        for (int i = lane; i < K * K; i += WAVE) s_A[(i % K) * K + (i / K)] = src[i];   // element (r, c) at r * K + c
        wave_sync();
    };
    load();
    double b = lane < K ? bvec[(size_t)sidx * K + lane] : 0.0;
    int rt = 0;
    if (!chol_solve_lds(s_A, K, K, b, lane)) {
        wave_sync();
        load();
        b = lane < K ? bvec[(size_t)sidx * K + lane] : 0.0;
        rt = lu_solve_lds(s_A, K, K, b, lane) ? 1 : -1;
    }
    if (lane < K) x[(size_t)sidx * K + lane] = b;
    if (lane == 0 && route) route[sidx] = rt;
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

// S[j][off_i + l] = sum_{r in level l of covariate i} x_rj    (fixed member order); with `codes`: train entries only
__global__ void __launch_bounds__(256) k_level_sums(const double *__restrict__ vals, const uint8_t *__restrict__ codes,
                                                    int64_t pitch, int p,
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
    for (int m = ptr[l]; m < ptr[l + 1]; ++m) {
        const size_t a = (size_t)j * pitch + mem[m];
        if (!codes || (codes[a] & CODE_TRAIN)) s += vals[a];
    }
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

// the same for sample r against the gene-major X: out[r][k] = sum_j X[j][r] F[j][k]
__global__ void __launch_bounds__(64) k_row_dense_xty(const double *__restrict__ X, int64_t pitch, int len,
                                                      const double *__restrict__ F, int K, int KP, double *__restrict__ out)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    for (int k = 0; k < KP; ++k) {
        double s = 0.0;
        if (k < K)
            for (int j = lane; j < len; j += 64) s += X[(size_t)j * pitch + r] * F[(size_t)j * KP + k];
        s = wave_sum(s);
        if (lane == 0) out[(size_t)r * KP + k] = s;
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
    if (lane < K) q_out[(size_t)u * K + lane] = qfull[(size_t)u * KP + lane] - s_H[(KP - 1) * KP + lane];
}

}  // namespace insider

#include "insider_cd_row16.hpp"
#include "insider_cd_reg.hpp"
#include "insider_ridge_reg.hpp"
#include "insider_mm.hpp"
#include "insider_row_merged.hpp"
#include "insider_col_factored.hpp"
#include "insider_cont_v2.hpp"
