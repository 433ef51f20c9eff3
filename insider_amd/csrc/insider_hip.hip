// insider_hip.hip — host driver and C ABI of libinsider_hip.so (see include/insider_hip.h).
//
// The outer alternating loop is the reference's optimize() (src/optimize.cpp:255-422) re-derived as
// "statistics, then solve" (DESIGN.md section 2): two streaming passes over (X, mask codes) per outer iteration
// — one per sample for the row update, one per gene fused with the elastic-net solve — and a handful of small
// dense kernels.  Everything is enqueued on one HIP stream; the host synchronises only at the reference's loss
// checkpoints (every 10th iteration) and around the optional cross-rank all-reduce.
#include "insider_kernels.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/insider_hip.h"

using namespace insider;

namespace {

thread_local std::string g_err;
thread_local double g_last_cd_ms = 0.0;   // per calling thread: the ABI is re-entrant per handle / per thread

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIPCHECK(call)                                                                                  \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? INSIDER_ERR_ALLOC : INSIDER_ERR_HIP,                \
                        std::string(#call) + ": " + hipGetErrorString(e_));                             \
    } while (0)

#define KCHECK() HIPCHECK(hipGetLastError())

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

template <typename T>
int dmalloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHECK(hipMalloc((void **)p, count * sizeof(T)));
    return INSIDER_OK;
}

// device temporaries of the handle-less entry points: freed on every exit path
struct DevBufs {
    std::vector<void *> ptrs;
    std::vector<hipEvent_t> events;
    ~DevBufs()
    {
        for (void *q : ptrs) if (q) (void)hipFree(q);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
    template <typename T>
    int alloc(T **p, size_t count)
    {
        int rc = dmalloc(p, count);
        if (rc == INSIDER_OK) ptrs.push_back((void *)*p);
        return rc;
    }
};

struct CovTables {   // per covariate, device
    int L = 0, nchunks = 0;
    int *chunk_level = nullptr, *chunk_begin = nullptr, *chunk_end = nullptr, *lvl_chunk_ptr = nullptr;
    // merged masked row update (insider_row_merged.hpp), built once per data set
    uint32_t *grp = nullptr;          // [p][L + 1] positions of the level groups inside each gene's sorted held-out samples
    uint16_t *slev = nullptr;         // per entry of every gene's level-grouped held-out samples: stacked level of each other covariate
    uint32_t *item_begin = nullptr, *item_end = nullptr;   // weighted-SYRK work items: ranges of the (gene, count) lists
    int *lvl_item_ptr = nullptr;      // [L + 1] items of every level
    int nitems = 0;
    int64_t npairs = 0;               // (level, gene) pairs with held-out samples
    int *wl_idx = nullptr;            // (gene, count) lists of every level, padded like the held-out lists
    double *wl_w = nullptr;
    double *paircnt = nullptr;        // [L][SLcat] samples in (level of this covariate, stacked level of another one)
};
// continuous columns share one table: a single pseudo-level whose members are all samples, in 16-sample chunks


}  // namespace

struct insider_hip_handle {
    int device = 0;
    // The read-only DATA SET (X-derived lists, level sums, pair counts, chunk tables: everything insider_hip_create builds)
    // may be shared by several handles (insider_hip_clone): each has its own factor workspace, streams and options, so that
    // several fits of one data set — tune()'s grid points — run on the GPU at the same time.  The device arrays are freed
    // by the last handle that goes.
    std::atomic<int> *data_refs = nullptr;
    hipStream_t stream = nullptr;
    // scheduling work that nothing but the next column solve needs (sweep keys -> gene order, the next iteration's sweep-order
    // table) runs on a side stream, next to the row update, between two events
    hipStream_t side = nullptr;
    hipEvent_t ev_cd_done = nullptr, ev_side_done = nullptr;
    bool side_pending = false;
    // the weighted SYRK of the merged row update depends on C only: all covariates' level sums are formed on a second
    // side stream while the main stream computes V, u and U'C
    hipStream_t side2 = nullptr;
    hipStream_t side3 = nullptr;      // C'C and (S^train C') of the merged row update, next to the weighted SYRK
    hipEvent_t ev_prep = nullptr;
    hipEvent_t ev_c_ready = nullptr;
    hipEvent_t ev_head = nullptr;     // recorded on side2 in front of the level Gram GEMM: the main chain's k_gene_u waits for it
    int row_head = 1;                 // option "row_head": that wait (1 = on)
    std::vector<hipEvent_t> ev_w;
    double *lvl_sum_all = nullptr;    // [SLcat][STAT + 2 KP + 2]: the level records of every covariate
    bool w_ready = false;
    double *gram_part2 = nullptr, *sc_part2 = nullptr;   // partial-sum buffers of the side-stream products
    hipEvent_t ev_a_ready = nullptr, ev_qfull = nullptr, ev_qheld = nullptr, ev_q_early = nullptr;
    int q_kb = 0;                     // rows [0, q_kb) of the stacked factors already sit in Qfull / Qheld (launch_q_early); 0: none
    bool qfull_pending = false;
    bool qheld_pending = false;       // Qheld = S^held A of the factored column statistics is being formed on side3 (phase_R)
    int64_t n = 0, p = 0, ldn = 0, ldp = 0;
    int c = 0, SL = 0, SLP = 0;   // SL: rows of the stacked row factors = all levels of all covariates + m
    int m = 0, SLcat = 0;          // continuous covariates (columns of ctns_confounder) and the categorical level total
    double *Zc = nullptr;          // m x n
    int *one_count = nullptr;      // a "member count" of 1 for the single pseudo-level of a continuous column
    std::vector<int> n_levels, lvl_off;   // lvl_off has c + 1 entries
    // data-set state (device)
    double *X = nullptr, *Xt = nullptr;
    uint8_t *codes = nullptr, *codes_t = nullptr;
    int *lev = nullptr, *lvl_off_d = nullptr, *members_all = nullptr, *lvl_ptr_all = nullptr, *lvl_count_all = nullptr;
    std::vector<CovTables> cov;
    CovTables cont;                // chunk tables shared by every continuous column
    // merged row update with continuous covariates (m <= 4, real-valued counts ColFacArgs::zt): column k is a ONE-level
    // covariate whose membership weights are z_rk — its (gene, weight) list carries sum_{r in H(j)} z_rk^2, its pair "counts"
    // sum_{r in l} z_rk per categorical level and (Z'Z)[k][k'] per other column, its |l| = sum_r z_rk^2
    std::vector<CovTables> contm;
    double *cont_cnt = nullptr;    // [m] sum_r z_rk^2
    bool cont_merged = false;
    int *ident_members = nullptr;  // 0..n-1
    int max_chunks = 0, max_L = 0;
    double *S = nullptr, *yy_train = nullptr, *yy_all = nullptr;
    double cnt_train = 0, cnt_test = 0;
    bool no_na = false;     // every entry is train or test: held-out == test
    // held-out lists (element index, value) of every gene (col) and of every sample (row); padded to LIST_ALIGN
    uint32_t *col_ptr = nullptr, *row_ptr = nullptr;
    int *col_idx = nullptr, *row_idx = nullptr;
    double *col_val = nullptr, *row_val = nullptr;
    uint8_t *col_flag = nullptr;   // per column-side list entry: 1 = test entry (only kept when the data has NA entries)
    uint64_t col_entries = 0, row_entries = 0;
    // factor-dependent workspace for the current K
    int K = 0, NB = 0, KP = 0, nseg = 1, seg_len = 0;
    double *Astack = nullptr, *R = nullptr, *C = nullptr, *RtR = nullptr, *CCt = nullptr, *Qfull = nullptr, *SC = nullptr;
    double *stat = nullptr, *stat_col = nullptr, *gram_part = nullptr, *sc_part = nullptr, *lvl_part = nullptr, *eq = nullptr;
    double *lvl_sum = nullptr;
    double *fperm = nullptr;      // max(n, p) x KP: the factor rows in k_tile_perm's order (k_list_stats4)
    double *lvl_zero = nullptr;   // max_L x (STAT + 2 KP + 2) zeros: the (empty) held-out sums of the unmasked row update (unmasked_fused)
    double *Strain = nullptr;         // per-level sums of X over TRAIN entries (p x SLP), once per data set
    double *Sheld = nullptr;          // S - Strain: per-level sums over the held-out entries
    double *Qheld = nullptr;          // p x KP workspace: sum_l A_l' Sheld[j][l]
    int col_factored = 1;             // option: factored column statistics (insider_col_factored.hpp): 0 list kernel, 1 cost model, 2 look-up form, 3 pair-count form
    uint8_t *cf_cnt = nullptr;        // dense pair counts of every gene (pair-count form), static per data set
    float *cf_hn = nullptr;           // 1/2 held-out count per (gene, level) in the pair-count kernel's order (ColFacArgs::hn)
    double *cf_zt = nullptr;          // real-valued counts of the continuous covariates in the same order (ColFacArgs::zt), m <= 4
    bool cf_pair_ok = false;
    int cf_pos[CF_MAXC] = {0};        // position of covariate i in cf's order (decreasing level count)
    int row_counts = 1;               // option: k_gene_u from the dense pair counts when they exist
    ColFacArgs cf;                    // its static part (filled at create)
    size_t cf_lds = 0;
    double *U = nullptr, *Ylvl = nullptr, *wpart = nullptr, *Vlev = nullptr;   // merged row update workspace
    int max_items = 0;
    bool merged = false;              // the merged masked row update is available (categorical covariates only)
    int row_merged = 1;               // option: use it
    int row_gemm = 1;                 // option "row_gemm": weighted SYRK of a many-level covariate as one GEMM over genes (k_wgemm)
    double *wg_part = nullptr;        // its per-slab partial sums
    int wg_waves = 1024;              // option "row_gemm_waves": waves the GEMM is cut into (sets the number of gene slabs)
    uint8_t *wg_pair = nullptr;       // packed pair index -> (a, b), a >= b: [2][16 ntile]
    int row_fused = 1;                // option "row_fused": level records' tail + equations + solve of the merged update in one launch
    double *sse_train = nullptr, *sse_test = nullptr, *b2 = nullptr, *b1 = nullptr, *loss_buf = nullptr, *stage = nullptr;
    int *sweeps = nullptr, *failflag = nullptr;
    int *sweep_key = nullptr;   // smoothed sweep counts: the longest-first scheduling key (k_sched_bucket)
    unsigned long long *sweep_total = nullptr;
    unsigned *pc4_ticket = nullptr;     // k_col_paircnt4's gene tickets (main launch, long-gene launch): counters that only grow
    unsigned pc4_base[4] = {0, 0, 0, 0};   // ... and the value each set stands at when its next launch starts: [2 site + (one counter ? 1 : 0)]
    // where the register-resident sweep kernel of the current K keeps its table of code blocks (K <= 32; 0 = not asked yet): the
    // order table holds absolute block addresses (insider_cd_reg.hpp), published by a probe launch of that kernel
    unsigned long long cd_code_base = 0, cd_pair_base = 0;
    unsigned long long *code_base_dev = nullptr;   // where the probe launch stores them (workspace)
    int q_split = 0;                   // option "q_split" (experiment of round 5, off: no gain — the early parts slow the memory-bound kernels of the last update by what they save): Qfull / Qheld in two parts, the first beside the last block of the row phase (launch_q_early)
    int join_lean = 0;                 // option "join_lean" (bits): 1 = ev_prep behind ev_w, 2 = ev_side_done behind ev_qfull (one stream join where two were), 4 = Qfull behind Qheld
    int mm_fast = 1;                   // option "mm_fast": the small dense products on k_mm_rows2 / k_mm_reduce2 (default; 2: two column tiles per wave in the reductions)
    int col_mfma4 = 1;                 // option "col_mfma4": pair-count statistics with the second product on v_mfma_f64_4x4x4 (k_col_paircnt4; default)
    int cd_pairs = 1;                  // option "cd_pairs": route the sweeps through the kernel's blocks of two coordinate steps (default)
    uint8_t *order = nullptr;          // the sweep-order table the next column solve reads: one of order_buf
    uint8_t *order_buf[2] = {nullptr, nullptr};   // two tables: the next outer iteration's is built while the current solve runs
    hipEvent_t ev_tab = nullptr;
    int order_rows = 0;
    // gene scheduling for the CD kernel: genes sorted by the sweep count of their previous solve
    int *gene_perm = nullptr;
    // the bucket sort behind it (k_sched_bucket / k_sched_scatter): two alternating sets of bucket counters, per gene its bucket
    // and its rank in the bucket
    int *sched_cnt[2] = {nullptr, nullptr}, *sched_rank = nullptr;
    uint16_t *sched_bkt = nullptr;
    int sched_flip = 0;
    // split column solves (steady-state outer iterations): the genes predicted longest — whole buckets of the launch order,
    // at most cd_long_frac of the genes — get their statistics and their solve on a stream of their own, ahead of everyone
    // else's statistics: the solve of the longest gene is the critical path of the column step (a sequential recurrence of
    // sweeps x K steps x ~44 ns), and it no longer waits for the statistics of the other genes
    int *sched_long = nullptr;        // device: {n_long, last long bucket} of the current gene_perm (k_sched_scatter)
    bool sched_long_valid = false;
    int n_simd = 1024;                // SIMDs of the device (4 per CU)
    int cd_split = 0;                 // option "cd_split"
    double cd_long_frac = 0.03;       // option "cd_long_frac"
    hipStream_t lng = nullptr;
    hipEvent_t ev_long_go = nullptr, ev_long_done = nullptr;
    bool long_pending = false;
    // multi-pass column solves in the cold outer iterations (CdParams::sweep_limit): saved state of the unfinished genes,
    // their estimated remaining lengths (two buffers, alternating between passes) and the order of the next pass
    double *cd_hsave = nullptr, *cd_isave = nullptr;
    uint32_t *cd_pass_slot = nullptr;
    int *cd_pass_perm[2] = {nullptr, nullptr}, *cd_pass_cnt = nullptr;   // cd_pass_cnt: CD_BUCKETS counters + 2 list lengths
    int list_fine = 1;   // option "list_fine": 1 (default) = k_list_stats4 (4x4x4 matrix instruction) where it applies, 0 = k_list_stats
    int cd_cold_iters = 3, cd_pass_first = 64, cd_pass_ratio = 4;   // options "cd_cold_iters", "cd_pass1" (0 = single pass), "cd_pass_ratio"
    // longest-first gene orders of outer iterations 0..2 of the previous optimize() on this handle: the early iterations
    // of the next call (tune()'s next grid point) have similar per-gene sweep counts, its later ones do not
    static constexpr int EARLY = 3;
    int *perm_early[EARLY] = {nullptr, nullptr, nullptr};
    bool have_early[EARLY] = {false, false, false};
    bool have_perm = false;
    size_t stage_count = 0;
    int gram_blocks_p = 0, gram_blocks_n = 0, sc_blocks = 0;
    // sharding
    int64_t gene_offset = 0;
    int rank = 0, world = 1;
    insider_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    ncclComm_t comm = nullptr;     // RCCL communicator over the gene-sharded ranks (insider_hip_comm_init); owned
    // options
    int max_sweeps = 1 << 24, order_mode = 0, profile = 0, verbose = 0, cd_variant = 0, force_allreduce = 0;
    // profile of the last optimize()
    std::vector<hipEvent_t> ev_col, ev_row, ev_cd, ev_test;
    double prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double steady_cd_ms = 0, steady_col_ms = 0;   // means over the outer iterations >= 5 of the last profiled optimize()
    // of the last optimize() / optimize_col(): genes whose elastic-net solve was ended by max_sweeps, not by convergence
    // (the reference has no cap, src/coordinate_descent.cpp:86-114), and the longest solve in sweeps
    int cap_hits = 0, max_gene_sweeps = 0;
};

namespace {

// events that order the handle's own streams against each other on ONE device: no timing, and no system-scope fence — what one
// stream's kernels wrote must reach the other stream's kernels (device scope: every kernel boundary does that), not the host
constexpr unsigned EV_SYNC = hipEventDisableTiming | hipEventDisableSystemFence;

constexpr int PC4_PARTS = 16;        // ticket counters of k_col_paircnt4 per launch site
constexpr int MM_FAST_MIN = 16384;   // rows from which k_mm_rows2 / k_mm_reduce2 run (below: the staging and the longer waves cost more than they save; c1: 5000 genes)
constexpr int MM_SLAB = 128;  // rows per partial of the reduction products (insider_mm.hpp)

// every K-dependent device buffer of a handle (ensure_workspace), as pointer slots
std::vector<void **> workspace_slots(insider_hip_handle *h)
{
    std::vector<void **> v;
    auto add = [&v](auto &ptr) { v.push_back(reinterpret_cast<void **>(&ptr)); };
    add(h->Astack); add(h->R); add(h->C); add(h->RtR); add(h->CCt); add(h->Qfull); add(h->SC); add(h->stat); add(h->stat_col);
    add(h->gram_part); add(h->sc_part); add(h->gram_part2); add(h->sc_part2); add(h->lvl_part); add(h->lvl_sum); add(h->lvl_zero); add(h->fperm); add(h->lvl_sum_all);
    add(h->U); add(h->Ylvl); add(h->wpart); add(h->Vlev); add(h->Qheld); add(h->eq); add(h->sse_train); add(h->sse_test); add(h->b2);
    add(h->b1); add(h->loss_buf); add(h->stage); add(h->wg_part); add(h->wg_pair); add(h->sweeps); add(h->sweep_key); add(h->failflag);
    add(h->sweep_total); add(h->pc4_ticket); add(h->order_buf[0]); add(h->order_buf[1]); add(h->gene_perm); add(h->sched_cnt[0]); add(h->sched_cnt[1]);
    add(h->sched_rank); add(h->sched_bkt); add(h->sched_long); add(h->cd_hsave); add(h->cd_isave); add(h->cd_pass_slot);
    add(h->cd_pass_perm[0]); add(h->cd_pass_perm[1]); add(h->cd_pass_cnt); add(h->code_base_dev);
    for (int e = 0; e < insider_hip_handle::EARLY; ++e) add(h->perm_early[e]);
    return v;
}

// drop the workspace WITHOUT freeing it (a clone starts from a copy of its source's fields: the buffers are the source's)
void forget_workspace(insider_hip_handle *h)
{
    for (void **slot : workspace_slots(h)) *slot = nullptr;
    h->order = nullptr;
    h->order_rows = 0;
    h->cd_code_base = h->cd_pair_base = 0;
    h->sched_long_valid = false;
    for (int e = 0; e < insider_hip_handle::EARLY; ++e) h->have_early[e] = false;
    h->have_perm = false;
    h->K = 0;
}

void free_workspace(insider_hip_handle *h)
{
    for (void **slot : workspace_slots(h))
        if (*slot) (void)hipFree(*slot);
    forget_workspace(h);
}

// The weighted SYRK of covariate i as a GEMM over genes (k_wgemm, insider_row_merged.hpp)?  Needs the static half-count table
// of the pair-count statistics; pays when the covariate has at least four tiles of 16 levels and the GEMM needs clearly fewer
// MFMAs than the per-(level, gene) form.
struct WgPlan {
    bool use = false;
    int tiles = 0, LT = 0, zch = 0, ntile = 0, npair = 0, slab = 0, nslab = 0;
};
WgPlan wgemm_plan(const insider_hip_handle *h, int i, int K, bool whatever_the_option = false)
{
    WgPlan w;
    if (!((h->row_gemm || whatever_the_option) && h->cf_pair_ok && h->cf_hn && h->merged && h->c <= CF_MAXC && K >= 1)) return w;
    const int NB = (K + 1 + 15) / 16;
    const int L = h->cov[i].L, NBLK = NB * (NB + 1) / 2;
    w.tiles = cdiv(L, 16);
    w.npair = K * (K + 1) / 2;
    w.ntile = cdiv(w.npair, 16);
    if (w.tiles < 4 || (double)w.tiles * w.ntile >= 0.9 * (double)L * NBLK) return w;
    w.zch = cdiv(w.tiles, 7);
    w.LT = cdiv(w.tiles, w.zch);
    const int waves_per_slab = cdiv(w.ntile, 2) * w.zch;
    // one wave per SIMD: the kernel runs beside the main stream's V -> u -> U'C chain (other waves fill the machine), its
    // operands are prefetched a step ahead, and every slab costs a partial record (levels x pairs doubles) to write and re-read
    // (rounded DOWN: at most wg_waves waves in all — with the default, one per SIMD: a surplus block would run as a second round)
    const int want = std::max(1, std::min<int>(h->wg_waves / waves_per_slab, (int)cdiv(h->p, 64)));
    w.slab = (int)round_up(cdiv(h->p, want), 4);
    w.nslab = (int)cdiv(h->p, w.slab);
    w.use = true;
    return w;
}

int ensure_workspace(insider_hip_handle *h, int K)
{
    if (K < 1 || K > INSIDER_MAX_K) return fail(INSIDER_ERR_UNSUPPORTED, "K must be in 1..63");
    if (h->K == K) return INSIDER_OK;
    free_workspace(h);
    const int NB = (K + 1 + 15) / 16, KP = 16 * NB, NBLK = NB * (NB + 1) / 2, STAT = NBLK * 256;
    h->NB = NB;
    h->KP = KP;
    // row-side segmentation: enough work items to fill 256 CUs even for few samples
    // enough work items to fill 256 CUs even for few samples: split long lists into up to 64 segments
    const int64_t avg_batches = (int64_t)(h->row_entries / (uint64_t)LIST_ALIGN / (uint64_t)std::max<int64_t>(h->n, 1));
    int nseg = (int)std::min<int64_t>(std::max<int64_t>(1, cdiv(8192, h->n)), std::max<int64_t>(1, avg_batches / 16));
    nseg = std::min(nseg, 64);
    h->nseg = nseg;
    h->seg_len = 0;
    h->gram_blocks_p = cdiv(h->p, MM_SLAB);
    h->gram_blocks_n = cdiv(h->n, MM_SLAB);
    h->sc_blocks = cdiv(h->p, MM_SLAB);
    int rc;
    // 16 rows of zero padding: the pair-count statistics kernel reads whole blocks of 16 levels without clamping
    if ((rc = dmalloc(&h->Astack, (size_t)(h->SL + 16) * KP))) return rc;
    HIPCHECK(hipMemset(h->Astack, 0, (size_t)(h->SL + 16) * KP * sizeof(double)));
    if ((rc = dmalloc(&h->R, (size_t)h->n * KP))) return rc;
    if ((rc = dmalloc(&h->C, (size_t)(std::max<int64_t>(h->p, h->ldp) + 4) * KP))) return rc;   // + 4 zero rows: k_wgemm reads whole steps of four genes
    if ((rc = dmalloc(&h->RtR, (size_t)KP * KP))) return rc;
    if ((rc = dmalloc(&h->CCt, (size_t)KP * KP))) return rc;
    if ((rc = dmalloc(&h->Qfull, (size_t)h->p * KP))) return rc;
    if ((rc = dmalloc(&h->SC, (size_t)h->SL * KP))) return rc;
    if ((rc = dmalloc(&h->stat, (size_t)nseg * h->n * STAT))) return rc;
    if ((rc = dmalloc(&h->stat_col, (size_t)h->p * STAT))) return rc;
    if ((rc = dmalloc(&h->gram_part, (size_t)std::max(h->gram_blocks_p, h->gram_blocks_n) * KP * KP))) return rc;
    if ((rc = dmalloc(&h->sc_part, (size_t)h->sc_blocks * h->SL * KP))) return rc;
    if ((rc = dmalloc(&h->lvl_part, (size_t)h->max_chunks * (STAT + 2 * KP + 2)))) return rc;
    if ((rc = dmalloc(&h->lvl_sum, (size_t)std::max(h->max_L, 1) * (STAT + 2 * KP + 2)))) return rc;
    if (NB == 2 && K >= 16)   // (the statistics kernels that read it exist for 16 <= K <= 31)
        if ((rc = dmalloc(&h->fperm, (size_t)std::max(h->n, h->p) * KP))) return rc;
    if ((rc = dmalloc(&h->lvl_zero, (size_t)std::max(h->max_L, 1) * (STAT + 2 * KP + 2)))) return rc;
    HIPCHECK(hipMemsetAsync(h->lvl_zero, 0, (size_t)std::max(h->max_L, 1) * (STAT + 2 * KP + 2) * sizeof(double), h->stream));
    if (h->merged) {
        // k_wgemm (weighted SYRK as a GEMM over genes): partial sums per gene slab, and the packed pair index -> (a, b) table
        size_t wg_len = 0;
        int wg_ntile = 0;
        for (int i = 0; i < h->c; ++i) {
            const WgPlan w = wgemm_plan(h, i, K, true);
            if (w.use) {
                wg_len = std::max(wg_len, (size_t)w.nslab * (16 * w.tiles) * (16 * w.ntile));
                wg_ntile = w.ntile;
            }
        }
        if (wg_len) {
            if ((rc = dmalloc(&h->wg_part, wg_len))) return rc;
            std::vector<uint8_t> ab((size_t)2 * 16 * wg_ntile, (uint8_t)(KP - 1));   // padded pairs: column KP - 1 of C, always zero
            int idx = 0;
            for (int a = 0; a < K; ++a)
                for (int b = 0; b <= a; ++b, ++idx) {
                    ab[idx] = (uint8_t)a;
                    ab[(size_t)16 * wg_ntile + idx] = (uint8_t)b;
                }
            if ((rc = dmalloc(&h->wg_pair, ab.size()))) return rc;
            HIPCHECK(hipMemcpy(h->wg_pair, ab.data(), ab.size(), hipMemcpyHostToDevice));
        }
        const int LP = (int)round_up(std::max(h->max_L, 1), 2);
        if ((rc = dmalloc(&h->U, (size_t)h->p * LP))) return rc;
        if ((rc = dmalloc(&h->Ylvl, (size_t)std::max(h->max_L, 1) * KP))) return rc;
        if ((rc = dmalloc(&h->wpart, (size_t)std::max(h->max_items, 1) * STAT))) return rc;
        if ((rc = dmalloc(&h->lvl_sum_all, (size_t)std::max(h->SL, 1) * (STAT + 2 * KP + 2)))) return rc;
        if ((rc = dmalloc(&h->gram_part2, (size_t)std::max(h->gram_blocks_p, h->gram_blocks_n) * KP * KP))) return rc;
        if ((rc = dmalloc(&h->sc_part2, (size_t)h->sc_blocks * h->SL * KP))) return rc;
        if ((rc = dmalloc(&h->Vlev, (size_t)h->p * h->SLP))) return rc;
        HIPCHECK(hipMemsetAsync(h->Vlev, 0, (size_t)h->p * h->SLP * sizeof(double), h->stream));   // (columns are filled as they are first needed)
        if ((rc = dmalloc(&h->Qheld, (size_t)h->p * KP))) return rc;
    }
    if ((rc = dmalloc(&h->eq, (size_t)h->max_L * (KP * KP + KP)))) return rc;
    if ((rc = dmalloc(&h->sse_train, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->sse_test, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->b2, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->b1, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->loss_buf, 8))) return rc;
    h->stage_count = (size_t)std::max<int64_t>(std::max<int64_t>(h->p, h->n), h->SL) * KP;
    if ((rc = dmalloc(&h->stage, h->stage_count))) return rc;
    if ((rc = dmalloc(&h->sweeps, (size_t)h->p))) return rc;
    // [0] a system was singular, [1] ridge genes wait for the general route, [2] genes stopped by max_sweeps, [3] longest solve
    if ((rc = dmalloc(&h->failflag, 4))) return rc;
    if ((rc = dmalloc(&h->sweep_total, 256))) return rc;
    if ((rc = dmalloc(&h->pc4_ticket, (size_t)2 * (PC4_PARTS + 1) * 32))) return rc;   // (a 128-byte line per counter)
    HIPCHECK(hipMemsetAsync(h->pc4_ticket, 0, (size_t)2 * (PC4_PARTS + 1) * 32 * sizeof(unsigned), h->stream));
    for (unsigned &b : h->pc4_base) b = 0;
    if ((rc = dmalloc(&h->gene_perm, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->sched_cnt[0], (size_t)SCHED_BUCKETS))) return rc;
    if ((rc = dmalloc(&h->sched_cnt[1], (size_t)SCHED_BUCKETS))) return rc;
    if ((rc = dmalloc(&h->sched_rank, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->sched_bkt, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->sched_long, 2))) return rc;
    if ((rc = dmalloc(&h->sweep_key, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->cd_hsave, (size_t)h->p * KP))) return rc;
    if ((rc = dmalloc(&h->cd_isave, (size_t)h->p * KP))) return rc;
    if ((rc = dmalloc(&h->cd_pass_slot, (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->cd_pass_perm[0], (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->cd_pass_perm[1], (size_t)h->p))) return rc;
    if ((rc = dmalloc(&h->cd_pass_cnt, (size_t)CD_BUCKETS + 2))) return rc;
    if ((rc = dmalloc(&h->code_base_dev, 2))) return rc;
    for (int e = 0; e < insider_hip_handle::EARLY; ++e)
        if ((rc = dmalloc(&h->perm_early[e], (size_t)h->p))) return rc;
    HIPCHECK(hipMemsetAsync(h->sched_cnt[0], 0, SCHED_BUCKETS * sizeof(int), h->stream));
    HIPCHECK(hipMemsetAsync(h->sched_cnt[1], 0, SCHED_BUCKETS * sizeof(int), h->stream));
    HIPCHECK(hipMemsetAsync(h->sched_long, 0, 2 * sizeof(int), h->stream));
    h->sched_flip = 0;
    h->sched_long_valid = false;
    h->have_perm = false;
    // rows of the padded factor buffers beyond K must stay zero: C rows are gathered with pitch KP and the
    // pad genes of the transposed layout index rows p..ldp-1
    HIPCHECK(hipMemsetAsync(h->C, 0, (size_t)(std::max<int64_t>(h->p, h->ldp) + 4) * KP * sizeof(double), h->stream));
    HIPCHECK(hipMemsetAsync(h->R, 0, (size_t)h->n * KP * sizeof(double), h->stream));
    HIPCHECK(hipMemsetAsync(h->failflag, 0, 4 * sizeof(int), h->stream));
    h->K = K;
    return INSIDER_OK;
}

// ---- launch helpers (dispatch on NB) -----------------------------------------------------------------------
#define NB_DISPATCH(NBV, ...)                                                           \
    switch (NBV) {                                                                      \
        case 1: { constexpr int NB_ = 1; constexpr int WPB_ = 4; __VA_ARGS__; } break;   \
        case 2: { constexpr int NB_ = 2; constexpr int WPB_ = 4; __VA_ARGS__; } break;   \
        case 3: { constexpr int NB_ = 3; constexpr int WPB_ = 2; __VA_ARGS__; } break;   \
        default: { constexpr int NB_ = 4; constexpr int WPB_ = 1; __VA_ARGS__; } break;  \
    }

int launch_list_stats(insider_hip_handle *h, bool cols, int nseg, const double *F, double *stat,
                      const double *base = nullptr)
{
    const int units = cols ? (int)h->p : (int)h->n;
    const int64_t f_rows = cols ? h->n : h->p;
    const uint32_t *ptr = cols ? h->col_ptr : h->row_ptr;
    const int *lidx = cols ? h->col_idx : h->row_idx;
    const double *lval = cols ? h->col_val : h->row_val;
    const int64_t items = (int64_t)units * nseg;
    // 16 <= K <= 31: the 4x4x4 form of the matrix instruction (fewer wasted outputs: 28 tiles instead of 3 blocks at K = 25)
    const int NT = (h->K + 4) / 4;
    if (h->list_fine && h->NB == 2 && NT >= 5 && NT <= 8 && h->fperm) {
        // the rows in the tile-pair order the kernel's 16-byte loads want (a copy: every other consumer keeps F's order)
        hipLaunchKernelGGL(k_tile_perm, dim3(cdiv(f_rows * h->KP, 256)), dim3(256), 0, h->stream, F, f_rows, h->KP, h->fperm);
#define LS4(NT_)                                                                                                               \
    hipLaunchKernelGGL((k_list_stats4<2, NT_, 4>), dim3(cdiv(items, 4)), dim3(256), 0, h->stream, ptr, lidx, lval, units, nseg,   \
                       (const double *)h->fperm, f_rows, stat, base, h->K)
        switch (NT) {
            case 5: LS4(5); break;
            case 6: LS4(6); break;
            case 7: LS4(7); break;
            default: LS4(8); break;
        }
#undef LS4
        KCHECK();
        return INSIDER_OK;
    }
    NB_DISPATCH(h->NB, hipLaunchKernelGGL((k_list_stats<NB_, WPB_>), dim3(cdiv(items, WPB_)), dim3(WPB_ * 64), 0,
                                           h->stream, ptr, lidx, lval, units, nseg, F, f_rows, stat, base, h->K));
    KCHECK();
    return INSIDER_OK;
}

// ---- the small dense products on MFMA (insider_mm.hpp) -------------------------------------------------------------------
// k_mm_rows2: tiles of 16 rows one wave takes — one until the grid fills every SIMD twice, then as many as keep it at that
int mm_tiles_per_wave(const insider_hip_handle *h, int tiles) { return std::max(1, tiles / (2 * h->n_simd)); }

// out[M x KP] = X[M x Kd] W[Kd x KP]   (W row-major with pitch KP)
bool mm_rows2_fits(const insider_hip_handle *h, int64_t ldx, int M, int Kd)
{
    return h->mm_fast && M >= MM_FAST_MIN && ldx % 2 == 0 && (size_t)4 * cdiv(Kd, 16) * h->NB * 64 * sizeof(double) <= 64 * 1024;
}
// k_begin / k_end: a window of the inner dimension (columns of X, rows of W), k_begin even; accumulate: out += the window's
// product.  Only with k_mm_rows2 (mm_rows2_fits); the default is the whole product
int launch_mm_rows_kp(insider_hip_handle *h, const double *X, int64_t ldx, int M, int Kd, const double *W, double *out,
                      hipStream_t st = nullptr, int k_begin = 0, int k_end = -1, bool accumulate = false)
{
    if (!st) st = h->stream;
    if (k_end < 0) k_end = Kd;
    const bool window = k_begin != 0 || k_end != Kd || accumulate;
    const size_t lds2 = (size_t)4 * cdiv(k_end - k_begin, 16) * h->NB * 64 * sizeof(double);   // k_mm_rows2: W staged in LDS
    if (mm_rows2_fits(h, ldx, M, Kd) && k_begin % 2 == 0) {
        const int tiles = cdiv(M, 16), tpw = mm_tiles_per_wave(h, tiles);
        const int kread = (int)((ldx - k_begin) & ~(int64_t)1);
        NB_DISPATCH(h->NB, {
            (void)WPB_;
            hipLaunchKernelGGL((k_mm_rows2<NB_, false>), dim3(cdiv(cdiv(tiles, tpw), 4), 1), dim3(256), lds2, st, X + k_begin, ldx, M,
                               k_end - k_begin, W + (size_t)k_begin * h->KP, h->KP, h->KP, out, (int64_t)h->KP, h->KP, tpw, kread,
                               accumulate ? 1 : 0);
        });
        KCHECK();
        return INSIDER_OK;
    }
    if (window) return fail(INSIDER_ERR_ARG, "a window of the product needs k_mm_rows2");
    NB_DISPATCH(h->NB, {
        (void)WPB_;
        hipLaunchKernelGGL((k_mm_rows<NB_, false>), dim3(cdiv(cdiv(M, 16), 4), 1), dim3(256), 0, st, X, ldx, M, Kd, W,
                           h->KP, h->KP, out, (int64_t)h->KP, h->KP);
    });
    KCHECK();
    return INSIDER_OK;
}

// part[slab][L][KP] = sum over slabs of MM_SLAB rows of X[m][l] Y[m][n], then the fixed-order sum over slabs -> out[L][KP]
// out == nullptr: the partial sums are left in `part` for the consumer to add up (k_level_merged); returns the slab count in *nslab
int launch_mm_reduce_kp(insider_hip_handle *h, const double *X, int64_t ldx, const double *Y, int M, int L, double *part,
                        double *out, hipStream_t st = nullptr, int *nslab = nullptr)
{
    if (!st) st = h->stream;
    const int slabs = cdiv(M, MM_SLAB);
    if (nslab) *nslab = slabs;
    if (h->mm_fast && M >= MM_FAST_MIN && L > 16) {   // several column tiles of X per wave, deeper look-ahead; the same partial sums (k_mm_reduce2)
        const int lt = cdiv(L, 16);
#define MR2(NBV, LTV)                                                                                                         \
    hipLaunchKernelGGL((k_mm_reduce2<NBV, LTV>), dim3(slabs, cdiv(lt, LTV)), dim3(64), 0, st, X, ldx, Y, (int64_t)h->KP, M,   \
                       MM_SLAB, L, h->KP, part, h->KP)
        NB_DISPATCH(h->NB, {
            (void)WPB_;
            if (lt <= 2 || h->mm_fast == 2 || NB_ > 2) MR2(NB_, 2);
            else MR2(NB_, 4);
        });
#undef MR2
    } else
    NB_DISPATCH(h->NB, {
        (void)WPB_;
        hipLaunchKernelGGL((k_mm_reduce<NB_>), dim3(slabs, cdiv(L, 16)), dim3(64), 0, st, X, ldx, Y, (int64_t)h->KP, M,
                           MM_SLAB, L, h->KP, part, h->KP);
    });
    KCHECK();
    if (out)
        hipLaunchKernelGGL(k_sum_partials, dim3(cdiv(L * h->KP, 16)), dim3(256), 0, st, (const double *)part, slabs,
                           L * h->KP, out);
    KCHECK();
    return INSIDER_OK;
}

// the row16 kernel with three slots keeps up to 4 x 48 x 48 doubles of Gram matrices per wave in LDS (72 KB): beyond the 64 KB a
// kernel may ask for dynamically without opting in
int r16_wide_lds(size_t bytes)
{
    if (bytes > 160 * 1024) return fail(INSIDER_ERR_UNSUPPORTED, "K too large for the LDS-resident sweep kernel");
    // the attribute belongs to the CURRENT device's function object: one flag per device (a process may drive several GPUs)
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (dev >= 64 || !(done.load() & bit)) {
        const int lim = 160 * 1024;
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cd_cols_r16<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cd_batch_r16<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
        done.fetch_or(bit);
    }
    return INSIDER_OK;
}

int launch_gram(insider_hip_handle *h, const double *F, int64_t rows, double *out, hipStream_t st = nullptr,
                double *part = nullptr)
{
    return launch_mm_reduce_kp(h, F, h->KP, F, (int)rows, h->KP, part ? part : h->gram_part, out, st);
}

int launch_build_R(insider_hip_handle *h)
{
    hipLaunchKernelGGL(k_build_R, dim3(cdiv(h->n * h->KP, 256)), dim3(256), 0, h->stream, (const int *)h->lev,
                       (const int *)h->lvl_off_d, h->c, (int)h->n, (const double *)h->Astack, h->KP,
                       (const double *)h->Zc, h->m, h->SLcat, h->R);
    KCHECK();
    return INSIDER_OK;
}

// R, R'R and Qfull from the current row factors (src/optimize.cpp:365-369 and the Xty of :222,235 via level sums)
// use_side: Qfull, which only the column solve reads, is formed on the side stream next to R'R and the column statistics
// r_is_current: the row updates have just rebuilt R (every row_update() ends with k_build_R): do not build it again
bool use_col_factored(const insider_hip_handle *h);
int phase_R(insider_hip_handle *h, bool use_side = false, bool r_is_current = false, bool want_qheld = false)
{
    if (use_side) {
        HIPCHECK(hipEventRecord(h->ev_a_ready, h->stream));
        // Qheld = S^held A, which the factored column statistics read: on the third stream (idle since the row phase's C'C),
        // beside R'R on the main one instead of behind it, and beside Qfull on the side stream.  (join_lean bit 4 puts Qfull
        // behind Qheld — alone, Qheld takes 22 instead of 33 us and the statistics start 11 us earlier — but Qfull then runs
        // beside the statistics kernel and costs it 40 us of LDS and matrix time for its own 21: measured, round 5.)
        // (launch_q_early: the rows [0, kb) of the stacked factors, final since the second-to-last update of the row phase, are
        // already in both products — formed beside the last update; only the last block's columns are left)
        const int kb = h->q_kb;
        h->q_kb = 0;
        HIPCHECK(hipStreamWaitEvent(h->side, h->ev_a_ready, 0));
        if (want_qheld && h->Qheld && use_col_factored(h)) {
            HIPCHECK(hipStreamWaitEvent(h->side3, h->ev_a_ready, 0));
            if (int rh = launch_mm_rows_kp(h, h->Sheld, h->SLP, (int)h->p, h->SL, h->Astack, h->Qheld, h->side3, kb, h->SL, kb > 0)) return rh;
            HIPCHECK(hipEventRecord(h->ev_qheld, h->side3));
            h->qheld_pending = true;
            if (h->join_lean & 4) HIPCHECK(hipStreamWaitEvent(h->side, h->ev_qheld, 0));
        }
        int rq = launch_mm_rows_kp(h, h->S, h->SLP, (int)h->p, h->SL, h->Astack, h->Qfull, h->side, kb, h->SL, kb > 0);
        if (rq) return rq;
        HIPCHECK(hipEventRecord(h->ev_qfull, h->side));
        h->qfull_pending = true;
    }
    int rc = r_is_current ? INSIDER_OK : launch_build_R(h);
    if (rc) return rc;
    rc = launch_gram(h, h->R, h->n, h->RtR);
    if (rc) return rc;
    if (use_side) return INSIDER_OK;
    return launch_mm_rows_kp(h, h->S, h->SLP, (int)h->p, h->SL, h->Astack, h->Qfull);
}

// Qfull = S A and Qheld = S^held A by parts (option "q_split"): the product over the rows [0, kb) of the stacked factors —
// every block of the row phase but its last — is formed on the side streams as soon as those rows are final, beside the last
// block's update (small dependent kernels that leave the machine idle); phase_R then adds the last block's columns.  At c3
// (100 + 10 levels) that leaves 10 of 110 columns behind the row phase: 8 instead of 33 us in front of the column statistics.
// Sums over the stacked levels in two runs [0, kb), [kb, SL) instead of one: agrees with the one-piece product to rounding.
int q_split_boundary(const insider_hip_handle *h, int masked, bool cont_follow)
{
    if (!h->q_split || !masked || !h->Qheld || !use_col_factored(h)) return 0;
    const int kb = cont_follow ? h->SLcat : (h->c >= 2 ? h->lvl_off[h->c - 1] : 0);
    if (kb < 16 || kb % 2 != 0 || kb >= h->SL || !mm_rows2_fits(h, h->SLP, (int)h->p, h->SL)) return 0;
    return kb;
}
int launch_q_early(insider_hip_handle *h, int kb)
{
    HIPCHECK(hipEventRecord(h->ev_q_early, h->stream));
    HIPCHECK(hipStreamWaitEvent(h->side3, h->ev_q_early, 0));
    if (int rh = launch_mm_rows_kp(h, h->Sheld, h->SLP, (int)h->p, h->SL, h->Astack, h->Qheld, h->side3, 0, kb, false)) return rh;
    HIPCHECK(hipStreamWaitEvent(h->side, h->ev_q_early, 0));
    if (int rq = launch_mm_rows_kp(h, h->S, h->SLP, (int)h->p, h->SL, h->Astack, h->Qfull, h->side, 0, kb, false)) return rq;
    h->q_kb = kb;
    return INSIDER_OK;
}

struct Timer {   // HIP-event pair around one launch on the library's stream (option "profile")
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int begin(insider_hip_handle *h, bool on)
    {
        if (!on || !h->profile) return INSIDER_OK;
        // timing only: no system-scope fence (the cache write-back and invalidation it brings cost the FOLLOWING kernel ~20 us behind
        // a statistics launch that has just written 300 MB; nothing reads these events' work from the host)
        HIPCHECK(hipEventCreateWithFlags(&e0, hipEventDisableSystemFence));
        HIPCHECK(hipEventCreateWithFlags(&e1, hipEventDisableSystemFence));
        HIPCHECK(hipEventRecord(e0, h->stream));
        return INSIDER_OK;
    }
    int end(insider_hip_handle *h, std::vector<hipEvent_t> &into)
    {
        if (!e0) return INSIDER_OK;
        HIPCHECK(hipEventRecord(e1, h->stream));
        into.push_back(e0);
        into.push_back(e1);
        return INSIDER_OK;
    }
};

// Builds the sweep-order table of outer iteration `iter` into order_buf[slot] (on `stream`); the caller makes it current
// (h->order) when its solve is launched.  Two buffers: an outer iteration's table depends on (seed, iter) only, so the NEXT one
// is built while the current solve runs — the sweep kernel leaves no room for other waves, so the builder runs in its tail,
// on SIMDs that have already drained — instead of competing with the row phase.
// 32 < K <= 48 with an l1 term: the register-resident kernel with its third slot's matrix columns in LDS (insider_cd_reg.hpp)
static bool reg3_path(int K, double la, int variant) { return K > 32 && K <= 48 && la > 0.0 && variant == 0; }

// the address of the table of code blocks of k_cd_cols_reg<., KMAX(K), true> on this device (K <= 32): one probe launch per workspace
int ensure_code_base(insider_hip_handle *h, int K)
{
    if (!reg_pairs(reg_kmax(K)) || h->cd_code_base) return INSIDER_OK;
    unsigned long long *d = h->code_base_dev;
    HIPCHECK(hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), h->stream));
    ColArgs a{};
    a.p = 0;
    a.K = K;
    a.KP = h->KP;
    a.code_base = d;
    REG_DISPATCH(K, hipLaunchKernelGGL((k_cd_cols_reg<SL_, KM_, true>), dim3(1), dim3(64), 0, h->stream, a));
    KCHECK();
    unsigned long long v[2] = {0, 0};
    HIPCHECK(hipMemcpyAsync(v, d, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (!v[0] || !v[1]) return fail(INSIDER_ERR_HIP, "the sweep kernel did not publish the addresses of its code blocks");
    h->cd_code_base = v[0];
    h->cd_pair_base = v[1];
    return INSIDER_OK;
}

int ensure_order_table(insider_hip_handle *h, uint64_t seed, uint32_t iter, int K, int max_sweeps, int order_mode, double la,
                       hipStream_t stream = nullptr, int slot = 0)
{
    if (int rb = ensure_code_base(h, K)) return rb;
    if (!stream) stream = h->stream;
    // one period of the order sequence at most (include/insider_perm.h): the table does not grow with max_sweeps
    const int rows = std::min<int64_t>(max_sweeps, INSIDER_PERM_PERIOD);
    if (h->order_rows < rows) {
        for (auto &b : h->order_buf) {
            if (b) (void)hipFree(b);
            b = nullptr;
            int rc = dmalloc(&b, (size_t)(rows + 4) * ORDER_ROW);   // + the look-ahead row (and the prologue's touch of the one after)
            if (rc) return rc;
        }
        h->order_rows = rows;
    }
    // rows for K > 32 carry 64 row offsets (row16 kernel) unless the solve takes the register-resident kernel's successor list
    hipLaunchKernelGGL(k_order_table, dim3(cdiv((int64_t)(rows + 1) * 64, 256)), dim3(256), 0, stream, seed, iter, K, rows,
                       order_mode, K * 8, reg_kmax(K), (K > 32 && !reg3_path(K, la, h->cd_variant)) ? 1 : 0, h->cd_code_base,
                       h->cd_pairs ? h->cd_pair_base : 0ull, h->order_buf[slot]);
    KCHECK();
    if (!h->order) h->order = h->order_buf[slot];
    return INSIDER_OK;
}

// most genes a split solve treats as long (whole buckets of the launch order up to this many)
int long_cap(const insider_hip_handle *h)
{
    const double f = h->cd_long_frac < 0.0 ? 0.0 : (h->cd_long_frac > 0.25 ? 0.25 : h->cd_long_frac);
    return (int)(f * (double)h->p);
}

// the launch order of the next column solve: genes by decreasing key, a bucket sort on a log scale (insider_kernels.hpp).
// sweeps != null: the keys are first updated from the last solve's sweep counts (reset: replaced, else smoothed)
// bucketed: the solve kernel has already done k_sched_bucket's part (ColArgs::sched_key ...), with the same counter set
int launch_gene_order(insider_hip_handle *h, const int *sweeps, int reset, int float_bits, hipStream_t st, bool bucketed = false)
{
    int *cnt = h->sched_cnt[h->sched_flip], *cnt_next = h->sched_cnt[h->sched_flip ^ 1];
    h->sched_flip ^= 1;
    if (!bucketed)
        hipLaunchKernelGGL(k_sched_bucket, dim3(cdiv(h->p, 256)), dim3(256), 0, st, sweeps, (int)h->p, reset, float_bits,
                           h->sweep_key, cnt, h->sched_bkt, h->sched_rank);
    KCHECK();
    hipLaunchKernelGGL(k_sched_scatter, dim3(cdiv(h->p, 256)), dim3(256), 0, st, (const int *)cnt, cnt_next,
                       (const uint16_t *)h->sched_bkt, (const int *)h->sched_rank, (int)h->p, h->gene_perm, long_cap(h),
                       h->sched_long);
    KCHECK();
    h->sched_long_valid = !float_bits;   // the sum-of-squares order of a first solve predicts no lengths
    return INSIDER_OK;
}

// Which kernel forms the column-side statistics: 0 = k_list_stats (one rank-one MFMA group per held-out entry), 1 =
// k_col_factored (look-up form), 2 = k_col_paircnt (pair-count form).  Issue-cycle models per gene with E held-out
// entries: the list kernel spends NBLK MFMAs (64 cycles) per 4 entries; the look-up form NB^2 MFMAs per 4 levels plus, per
// later covariate and 16-entry batch of a level group, 16 x (2 + NB) vector instructions (4.6 cycles); the pair-count form
// NB^2 MFMAs per 4 levels plus ceil(rows / 4) x NB MFMAs per 16 levels.  Measured at c3 / c5: look-up 0.51 vs list 1.30 ms
// (model 14.5k vs 48k cycles) and 2.44 vs 0.66 ms (72k vs 24k).
int col_stats_path(const insider_hip_handle *h)
{
    if (!(h->merged && h->col_factored && h->c <= CF_MAXC)) return 0;
    if (h->m > 0) return (h->cf_pair_ok && h->cf_zt) ? 2 : 0;   // continuous covariates: the pair-count form with real-valued counts, or the lists
    const bool lookup_fits =
        ((size_t)(h->cf.tab_rows + 1) * h->KP + 4 * 16 * 17) * sizeof(double) + (size_t)4 * CF_CAP * 2 <= 64 * 1024;
    if (h->col_factored == 3 && h->cf_pair_ok) return 2;                   // forced
    if (h->col_factored >= 2) return lookup_fits ? 1 : (h->cf_pair_ok ? 2 : 0);   // forced (3 without a count table: look-up form)
    const double E = (double)h->col_entries / (double)std::max<int64_t>(h->p, 1);
    const int NB = h->NB;
    const double list = E * (NB * (NB + 1) / 2) * 16.0;
    double fac = 0.0, pair = 0.0;
    for (int t = 0; t < h->cf.c; ++t) {
        const double batches = std::ceil(std::max(1.0, E / h->cf.L[t]) / 16.0);
        fac += std::ceil(h->cf.L[t] / 4.0) * (NB * NB * 64.0 + h->cf.nlater[t] * batches * 16.0 * (2 + NB) * 4.6);
        pair += std::ceil(h->cf.L[t] / 4.0) * NB * NB * 64.0 +
                std::ceil(h->cf.L[t] / 16.0) * ((h->cf.nlater[t] > 0 ? h->cf.nsteps * NB * 64.0 : 0.0) + 150.0);
    }
    double best = list;
    int path = 0;
    if (lookup_fits && 1.3 * fac < best) { best = 1.3 * fac; path = 1; }
    if (h->cf_pair_ok && 1.3 * pair < best) { best = 1.3 * pair; path = 2; }
    return path;
}
bool use_col_factored(const insider_hip_handle *h) { return col_stats_path(h) != 0; }

// the pair-count statistics kernel (insider_col_factored.hpp) on `blocks` blocks of four genes
int launch_paircnt(insider_hip_handle *h, const ColFacArgs &a, int blocks, hipStream_t st)
{
    if (h->col_mfma4 && h->NB <= 2 && !(a.zt && a.nsteps > 4)) {   // (real-valued counts with more than four k-steps: 180 registers, two waves per SIMD)
        // second product on the 4x4x4 matrix instruction, factor rows of every position staged in LDS (k_col_paircnt4);
        size_t quads = 1;
        for (int t = 0; t < a.c + (a.zt ? 1 : 0); ++t) quads += (size_t)(a.L[t] + 3) / 4;
        const size_t lds = ((size_t)4 * 16 * 17 + (size_t)h->KP * h->KP + (size_t)4 * a.nsteps * h->KP + 4 * quads * h->KP) * sizeof(double);
        if (lds <= 64 * 1024) {
            // as many blocks as stay resident (48.6 KB of LDS at c3: three per CU); each walks the groups of four genes with the grid's stride
            const int resident = std::max(1, std::min((int)(160 * 1024 / lds), (a.zt && a.nsteps > 4) ? 2 : 3)) * std::max(1, h->n_simd / 4);   // registers: 148 - 166 (180 with real-valued counts and more than four k-steps)
            int nb = std::min(blocks, resident);
            const int npart = nb >= PC4_PARTS ? PC4_PARTS : 1;      // ticket counters in use (k_col_paircnt4)
            nb -= nb % npart;
            const int nitems = a.list ? 4 * blocks : a.p;           // (a list launch: its bound; the kernel stops at *list_count)
            const int cap = cdiv(nitems, npart);
            // the two launch sites may run at the same time: a set of counters each; a launch with ONE counter (few blocks) has
            // a counter of its own behind the sixteen, so that the counters of a set always stand at the same value
            const int which = (st == h->lng ? 2 : 0) + (npart == 1 ? 1 : 0);
            unsigned *tk = h->pc4_ticket + ((size_t)(which >> 1) * (PC4_PARTS + 1) + (npart == 1 ? PC4_PARTS : 0)) * 32;
            const unsigned tbase = h->pc4_base[which];
#define PC4(NBV, MS)                                                                                                         \
    {                                                                                                                        \
        if (a.zt) hipLaunchKernelGGL((k_col_paircnt4<NBV, 4, MS, true>), dim3(nb), dim3(256), lds, st, a, nitems, tk, tbase, npart, cap); \
        else hipLaunchKernelGGL((k_col_paircnt4<NBV, 4, MS, false>), dim3(nb), dim3(256), lds, st, a, nitems, tk, tbase, npart, cap); \
    }
            if (h->NB == 1 && a.nsteps <= 4) PC4(1, 4)
            else if (h->NB == 1) PC4(1, 8)
            else if (a.nsteps <= 4) PC4(2, 4)
            else PC4(2, 8)
#undef PC4
            KCHECK();
            // what the launch takes from each counter in use: its items and one ticket per wave (only once it is known to be enqueued)
            h->pc4_base[which] += (unsigned)cap + 4u * (unsigned)(nb / npart);
            return INSIDER_OK;
        }
    }
    NB_DISPATCH(h->NB, {
        (void)WPB_;
        const size_t lds = ((size_t)4 * 16 * 17 + (size_t)Geo<NB_>::KP * Geo<NB_>::KP + (size_t)4 * a.nsteps * Geo<NB_>::KP) * sizeof(double);
        if (a.zt) hipLaunchKernelGGL((k_col_paircnt<NB_, 4, true>), dim3(blocks), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((k_col_paircnt<NB_, 4, false>), dim3(blocks), dim3(256), lds, st, a);
    });
    KCHECK();
    return INSIDER_OK;
}

// Should this outer iteration's column step run split (long genes on their own stream, ahead of the others' statistics)?
// Steady-state iterations only (the cold ones are throughput-bound and solve in passes), the pair-count statistics, the
// register-resident sweep kernel, and a launch order made from sweep counts.
bool use_split(const insider_hip_handle *h, int masked, double alpha, int outer_iter)
{
    if (!(h->cd_split && masked && alpha != 0.0 && h->cd_variant == 0 && h->K <= 32 && col_stats_path(h) == 2 &&
          outer_iter >= std::max(h->cd_cold_iters, (int)insider_hip_handle::EARLY) && h->have_perm && h->sched_long_valid &&
          long_cap(h) >= 4))
        return false;
    // Not on by default.  Measured on a 25000-gene slab of c4 (one rank of the 8-GPU configuration; tools/slab_trace.sh,
    // tools/tail_probe.py): the launch order predicts the tail well (the 50 longest genes of a solve are all among its first
    // 3 %, correlation of consecutive sweep counts 0.95), but the tail is BROAD, not a few outliers (median 171, p99 415, max
    // 612 sweeps; the longest gene outside the first 10 % still needs 425).  So the launch that holds everyone else is barely
    // shorter than the whole solve (0.69 - 0.72 ms against 0.73, which is what the longest gene takes alone), and the long
    // genes' sweeps next to the others' MFMA-bound statistics slow those down: 1.26 ms
    // per steady iteration unsplit, 1.30 / 1.33 / 1.40 ms with 3 / 10 / 25 % of the genes split off.
    return h->cd_split >= 2;   // forced
}

// masked Gram/XtY complement statistics of every gene (column side of src/optimize.cpp:216-222)
// split: the long genes' records first, on the stream `lng` (their solve follows there, launch_col_solve); the launch over
// all genes on the main stream skips them
int launch_col_stats(insider_hip_handle *h, bool timed, bool split = false)
{
    Timer t;
    int rc;
    if (use_col_factored(h) && h->qheld_pending) {   // Qheld = S^held A formed on side3 since the row factors were final (phase_R):
        HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_qheld, 0));   // joined BEFORE the timer, which then holds the statistics kernel alone
        h->qheld_pending = false;
        if ((rc = t.begin(h, timed))) return rc;
    } else {
        if ((rc = t.begin(h, timed))) return rc;
        if (use_col_factored(h))
            if ((rc = launch_mm_rows_kp(h, h->Sheld, h->SLP, (int)h->p, h->SL, h->Astack, h->Qheld))) return rc;
    }
    if (use_col_factored(h)) {
        ColFacArgs a = h->cf;
        a.K = h->K;
        a.Astack = h->Astack;
        a.Qheld = h->Qheld;
        a.RtR = h->RtR;
        a.yy_all = h->yy_all;
        a.yy_train = h->yy_train;
        a.stat = h->stat_col;
        if (col_stats_path(h) == 2) {
            a.cnt = h->cf_cnt;
            a.hn = h->cf_hn;
            a.zt = h->cf_zt;
            if (split) {
                // the long branch starts here: everything the main stream has produced so far (row factors, R'R, Qheld)
                // plus what the side stream prepares for the solve (launch order, sweep-order table, Qfull)
                HIPCHECK(hipEventRecord(h->ev_long_go, h->stream));
                HIPCHECK(hipStreamWaitEvent(h->lng, h->ev_long_go, 0));
                if (h->side_pending) HIPCHECK(hipStreamWaitEvent(h->lng, h->ev_side_done, 0));
                if (h->qfull_pending) HIPCHECK(hipStreamWaitEvent(h->lng, h->ev_qfull, 0));
                ColFacArgs al = a;
                al.list = h->gene_perm;
                al.list_count = h->sched_long;
                rc = launch_paircnt(h, al, cdiv(long_cap(h), 4), h->lng);
                if (rc) return rc;
                KCHECK();
                h->long_pending = true;
                a.skip_bkt = h->sched_bkt;
                a.skip_last = h->sched_long + 1;
                // the all-gene launch below reads sched_long / sched_bkt too, which k_sched_scatter wrote on the side stream:
                // the main stream must see them as well (it otherwise waits for the side stream only before the solve)
                if (h->side_pending) HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_side_done, 0));
            }
            rc = launch_paircnt(h, a, cdiv(h->p, 4), h->stream);
            if (rc) return rc;
        } else {
            NB_DISPATCH(h->NB, {
                (void)WPB_;
                const size_t lds = ((size_t)(a.tab_rows + 1) * Geo<NB_>::KP + (size_t)4 * 16 * 17) * sizeof(double) + (size_t)4 * CF_CAP * 2;
                hipLaunchKernelGGL((k_col_factored<NB_, 4>), dim3(cdiv(h->p, 4)), dim3(256), lds, h->stream, a);
            });
        }
        KCHECK();
    } else {
        rc = launch_list_stats(h, true, 1, h->R, h->stat_col, h->RtR);   // the record's K x K part = R'R - complement
        if (rc) return rc;
    }
    return t.end(h, h->ev_col);
}

// column update from the statistics: elastic-net CD (alpha > 0) or ridge (alpha == 0), or evaluation only
int launch_col_solve(insider_hip_handle *h, int masked, bool solve, double lambda, double alpha, double tol,
                     int checkpoint, bool timed, int outer_iter = -1, bool side = false)
{
    const bool early = outer_iter >= 0 && outer_iter < insider_hip_handle::EARLY;
    const int NBLK = h->NB * (h->NB + 1) / 2, STAT = NBLK * 256;
    // (every stream join is a barrier packet on the main queue, ~5 us of bubble each at c3: ev_qfull is recorded on the side stream
    // AFTER the previous iteration's ev_side_done — phase_R comes after side_close — so it stands for both)
    if (h->side_pending && !((h->join_lean & 2) && h->qfull_pending)) {   // the gene order / sweep-order table prepared on the side stream
        HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_side_done, 0));
    }
    h->side_pending = false;
    if (h->qfull_pending) {
        HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_qfull, 0));
        h->qfull_pending = false;
    }
    Timer t;
    int rc = t.begin(h, timed);
    if (rc) return rc;
    bool eval_after = false, fused_bucket = false;
    ColArgs eval_args;
    if (alpha == 0.0) {
        RidgeArgs a;
        a.stat = masked ? h->stat_col : nullptr;
        a.stat_len = STAT;
        a.p = (int)h->p;
        a.K = h->K;
        a.KP = h->KP;
        a.RtR = h->RtR;
        a.Qfull = h->Qfull;
        a.C = h->C;
        a.yy = masked ? h->yy_train : h->yy_all;
        a.lambda = lambda;
        a.solve = solve ? 1 : 0;
        a.checkpoint = checkpoint;
        a.sse_train = h->sse_train;
        a.b2 = h->b2;
        a.b1 = h->b1;
        a.sse_test = h->sse_test;
        a.test_from_stats = masked && h->no_na;
        a.fail = h->failflag;
        a.mark = h->sweeps;          // free in the alpha == 0 path: cleared below
        a.retry = h->failflag + 1;
        a.only_marked = 0;
        if (h->K <= 32 && h->cd_variant == 0) {
            if (solve) {
                HIPCHECK(hipMemsetAsync(h->sweeps, 0, (size_t)h->p * sizeof(int), h->stream));
                HIPCHECK(hipMemsetAsync(h->failflag + 1, 0, sizeof(int), h->stream));
            }
            REG_DISPATCH(h->K, hipLaunchKernelGGL((k_ridge_cols_reg<SL_, KM_>), dim3(cdiv(h->p, 4)), dim3(64), 0, h->stream, a));
            KCHECK();
            if (solve) {   // genes whose system was not positive definite: solve(..., likely_sympd)'s general route
                a.only_marked = 1;
                hipLaunchKernelGGL((k_ridge_cols<1>), dim3((unsigned)h->p), dim3(64), 0, h->stream, a);   // unmarked genes exit at once
            }
        } else {
            hipLaunchKernelGGL((k_ridge_cols<1>), dim3((unsigned)h->p), dim3(64), 0, h->stream, a);
        }
        KCHECK();
        if (solve) HIPCHECK(hipMemsetAsync(h->sweeps, 0, (size_t)h->p * sizeof(int), h->stream));
    } else {
        ColArgs a;
        a.stat = masked ? h->stat_col : nullptr;
        a.stat_len = STAT;
        a.p = (int)h->p;
        a.K = h->K;
        a.KP = h->KP;
        a.RtR = h->RtR;
        a.Qfull = h->Qfull;
        a.C = h->C;
        a.yy = masked ? h->yy_train : h->yy_all;
        a.mode = solve ? COL_CD : COL_EVAL;
        a.checkpoint = checkpoint;
        a.cd.lambda = lambda;
        a.cd.alpha = alpha;
        a.cd.tol = tol;
        a.cd.la = lambda * alpha;
        a.cd.l2 = lambda * (1.0 - alpha);
        a.cd.two_la = 2.0 * a.cd.la;
        a.cd.inv_two_la = a.cd.la > 0.0 ? 0.5 / a.cd.la : 0.0;
        a.cd.max_sweeps = h->max_sweeps;
        a.cd.order = h->order;
        a.sse_train = h->sse_train;
        a.b2 = h->b2;
        a.b1 = h->b1;
        a.sse_test = h->sse_test;
        a.test_from_stats = masked && h->no_na;
        a.sweeps = h->sweeps;
        a.sweep_bins = (timed && solve) ? h->sweep_total : nullptr;
        a.gene_perm = !solve ? nullptr : (early && h->have_early[outer_iter]) ? h->perm_early[outer_iter]
                                       : h->have_perm                         ? h->gene_perm
                                                                              : nullptr;
        a.hsave = h->cd_hsave;
        a.isave = h->cd_isave;
        a.pass_count = nullptr;
        a.slot_begin = nullptr;
        a.resume = 0;
        a.pass_slot = nullptr;
        a.bucket_cnt = nullptr;
        a.cap_hits = solve ? h->failflag + 2 : nullptr;
        a.sched_key = a.sched_cnt = a.sched_rank = nullptr;
        a.sched_bkt = nullptr;
        a.sched_reset = 0;
        const size_t r16_bytes = (size_t)r16_lds_doubles(h->K) * sizeof(double);
        // the register-resident kernel scales its state by 1 / (2 lambda alpha): lambda alpha = 0 (alpha < 0 or lambda = 0: no l1
        // term at all) takes the group kernel below
        if (h->cd_variant == 0 && (h->K <= 32 || reg3_path(h->K, a.cd.la, 0)) && a.cd.la > 0.0) {
            // Cold outer iterations: thousands of sweeps per gene whose counts no history predicts, so a wave's four genes
            // finish far apart (measured at c3: 1.17x / 1.44x / 2.1x the ideal wave time in outer iterations 0 / 1 / 2).
            // The solve then runs in passes over geometrically growing sweep ranges: a limited pass stops at its sweep
            // index, the genes still running save their state and an estimate of their remaining length (from the decay
            // of the loss change), k_pass_scatter groups them by that estimate, and the next pass continues them
            // bit-identically in waves of similar length (insider_cd_reg.hpp).  A pass with nothing left exits at once.
            if (solve) {   // every gene's part of the next launch order, when it finishes (launch_gene_order below skips k_sched_bucket)
                a.sched_key = h->sweep_key;
                a.sched_cnt = h->sched_cnt[h->sched_flip];
                a.sched_rank = h->sched_rank;
                a.sched_bkt = h->sched_bkt;
                a.sched_reset = (outer_iter < insider_hip_handle::EARLY || !h->have_perm) ? 1 : 0;
                fused_bucket = true;
            }
            int limits[16], npass = 0;
            if (solve && outer_iter >= 0 && outer_iter < h->cd_cold_iters && h->cd_pass_first >= 32)
                for (int64_t l = h->cd_pass_first; l < std::min<int64_t>(h->max_sweeps, 4 * (int64_t)INSIDER_PERM_PERIOD) && npass < 16;
                     l *= std::max(h->cd_pass_ratio, 2))
                    limits[npass++] = (int)l;   // the last pass runs from the last limit to the end, however far that is
            int start = 0;
            const int *perm_in = a.gene_perm;
            if (h->long_pending) {
                // split solve: the long genes (the first n_long slots of the launch order) on their own stream, right after their
                // statistics; everyone else here, from slot n_long on
                ColArgs al = a;
                al.cd.start_sweep = 0;
                al.cd.sweep_limit = 0;
                al.pass_count = h->sched_long;
                REG_DISPATCH(h->K, hipLaunchKernelGGL((k_cd_cols_reg<SL_, KM_, true>), dim3(cdiv(long_cap(h), 4)), dim3(64), 0, h->lng, al));
                KCHECK();
                HIPCHECK(hipEventRecord(h->ev_long_done, h->lng));
                a.slot_begin = h->sched_long;
            }
            for (int pass = 0;; ++pass) {
                const int limit = pass < npass ? limits[pass] : 0;
                a.cd.start_sweep = start;
                a.cd.sweep_limit = limit;
                a.pass_slot = npass ? h->cd_pass_slot : nullptr;
                a.bucket_cnt = limit ? h->cd_pass_cnt : nullptr;
                if (limit) HIPCHECK(hipMemsetAsync(h->cd_pass_cnt, 0, CD_BUCKETS * sizeof(int), h->stream));
                if (solve) { REG_ANY_DISPATCH(h->K, hipLaunchKernelGGL((k_cd_cols_reg<SL_, KM_, true>), dim3(cdiv(h->p, 4)), dim3(64), 0, h->stream, a)); }
                KCHECK();
                if (!limit) break;
                int *count_out = h->cd_pass_cnt + CD_BUCKETS + (pass & 1);
                hipLaunchKernelGGL(k_pass_scatter, dim3(cdiv(h->p, 256)), dim3(256), 0, h->stream,
                                   (const uint32_t *)h->cd_pass_slot, (const int *)h->cd_pass_cnt, perm_in, a.pass_count, (int)h->p,
                                   h->cd_pass_perm[pass & 1], count_out);
                KCHECK();
                perm_in = a.gene_perm = h->cd_pass_perm[pass & 1];
                a.pass_count = count_out;
                a.resume = 1;
                start = limit;
            }
            if (h->long_pending) {   // the column step ends when both parts have
                HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_long_done, 0));
                h->long_pending = false;
            }
            eval_after = checkpoint != 0;
            eval_args = a;
        } else if (h->cd_variant == 2 && h->K <= 16)
            hipLaunchKernelGGL((k_cd_cols_r16<1>), dim3(cdiv(h->p, 4)), dim3(64), r16_bytes, h->stream, a);
        else if (h->cd_variant == 2 && h->K <= 32)
            hipLaunchKernelGGL((k_cd_cols_r16<2>), dim3(cdiv(h->p, 4)), dim3(64), r16_bytes, h->stream, a);
        else if (h->K <= 16) hipLaunchKernelGGL((k_cd_cols<16, 4>), dim3(cdiv(h->p, 16)), dim3(256), 0, h->stream, a);
        else if (h->K <= 32) hipLaunchKernelGGL((k_cd_cols<32, 2>), dim3(cdiv(h->p, 4)), dim3(128), 0, h->stream, a);
        else if (h->cd_variant != 1 && h->K <= 48) {
            // 32 < K <= 48 when the register-resident kernel's three-slot form does not apply (cd_variant = 2, or no l1 term): four genes
            // per wavefront with the whole Gram matrices in LDS (row16 kernel, three coordinate slots per lane).  Beyond 48 a CU's LDS
            // holds one such wave and the kernel below is faster; it also stays as cd_variant = 1 (cross-check)
            if (int rl = r16_wide_lds(r16_bytes)) return rl;
            hipLaunchKernelGGL((k_cd_cols_r16<3>), dim3(cdiv(h->p, 4)), dim3(64), r16_bytes, h->stream, a);
        }
        else hipLaunchKernelGGL((k_cd_cols<64, 1>), dim3((unsigned)h->p), dim3(64), 0, h->stream, a);
        KCHECK();
    }
    if ((rc = t.end(h, h->ev_cd))) return rc;
    if (eval_after) {   // the per-gene loss statistics of the (updated) columns: the evaluation kernel, all genes (not part of the solve's time)
        eval_args.gene_perm = nullptr;
        eval_args.pass_count = nullptr;
        eval_args.slot_begin = nullptr;
        eval_args.resume = 0;
        if (h->K <= 32) {
            REG_DISPATCH(h->K, hipLaunchKernelGGL((k_cd_cols_reg<SL_, KM_, false>), dim3(cdiv(h->p, 4)), dim3(64), 0, h->stream, eval_args));
        } else {   // three slots: the row16 kernel's evaluation part (the sweep kernel's registers are all taken)
            const size_t eb = (size_t)r16_lds_doubles(h->K) * sizeof(double);
            if (int rl = r16_wide_lds(eb)) return rl;
            eval_args.mode = COL_EVAL;
            hipLaunchKernelGGL((k_cd_cols_r16<3>), dim3(cdiv(h->p, 4)), dim3(64), eb, h->stream, eval_args);
        }
        KCHECK();
    }
    if (solve && alpha != 0.0) {
        // schedule the next solve longest-first, genes of similar length sharing a wave.  (The bucket sort's order inside a
        // bucket is the order of the solve kernel's atomics: which genes share a wave — and so the timings — may differ from
        // run to run; no result depends on it.)
        hipStream_t st = h->stream;
        if (side) {
            HIPCHECK(hipEventRecord(h->ev_cd_done, h->stream));
            HIPCHECK(hipStreamWaitEvent(h->side, h->ev_cd_done, 0));
            st = h->side;
        }
        if ((rc = launch_gene_order(h, h->sweeps, (outer_iter < insider_hip_handle::EARLY || !h->have_perm) ? 1 : 0, 0, st, fused_bucket)))
            return rc;
        h->have_perm = true;
        if (early) {
            HIPCHECK(hipMemcpyAsync(h->perm_early[outer_iter], h->gene_perm, (size_t)h->p * sizeof(int),
                                    hipMemcpyDeviceToDevice, st));
            h->have_early[outer_iter] = true;
        }
        if (side) {   // the caller may add the next sweep-order table to the side stream, then closes it with side_close()
            h->side_pending = true;
        }
    }
    return INSIDER_OK;
}

int side_close(insider_hip_handle *h)
{
    if (h->side_pending) HIPCHECK(hipEventRecord(h->ev_side_done, h->side));
    return INSIDER_OK;
}

// sum over test entries of the squared residual per gene (evaluate(), src/utils.cpp:67); checkpoints only
int launch_test_sse(insider_hip_handle *h, int masked, bool timed)
{
    if (!masked) {
        HIPCHECK(hipMemsetAsync(h->sse_test, 0, (size_t)h->p * sizeof(double), h->stream));
        return INSIDER_OK;
    }
    if (h->no_na) return INSIDER_OK;   // the column kernel derived the test residuals from the statistics
    Timer t;
    int rc = t.begin(h, timed);
    if (rc) return rc;
    hipLaunchKernelGGL((k_test_sse_list<4>), dim3(cdiv(h->p, 4)), dim3(256), 0, h->stream, (const uint32_t *)h->col_ptr,
                       (const int *)h->col_idx, (const double *)h->col_val, (const uint8_t *)h->col_flag, (int)h->p,
                       (const double *)h->R, (const double *)h->C, h->K, h->KP, h->sse_test);
    KCHECK();
    return t.end(h, h->ev_test);
}

// masked update without per-sample statistics (insider_row_merged.hpp)?  Time model (us at c3 / c5 rates): per-sample
// path = one rank-one MFMA group per held-out entry + the level kernels; merged path per covariate = one weighted group per
// (level, gene) pair + one look-up per entry and other covariate + the small products over its levels.  Measured at
// c3: 0.55 vs 1.4 ms (model 0.45 vs 1.38); at c5 (4 covariates): 1.37 vs 0.85 ms (model 1.24 vs 0.84).
bool use_merged(const insider_hip_handle *h, int masked)
{
    if (!(masked && h->merged && h->row_merged && (h->m == 0 || h->cont_merged))) return false;
    if (h->row_merged == 2 || h->m > 0) return true;   // forced / continuous covariates: the per-sample pass is the slow alternative
    const int NB = h->NB ? h->NB : 2;
    const double mf = (NB * (NB + 1) / 2) * 16.0 / (1024.0 * 2100.0);   // us per rank-one group entry on the whole GPU
    const double E = (double)h->row_entries, pscale = (double)h->p / 5.0e4;
    const double old_us = E * mf * 1.15 + 50.0 * h->c;
    double merged_us = 0.0;
    for (int i = 0; i < h->c; ++i)
        merged_us += (double)h->cov[i].npairs * mf * 1.1 + E * (h->c - 1) * 1.3e-6 + 33.0 * pscale * h->SLcat / 110.0 / h->c +
                     37.0 * pscale * h->cov[i].L / 100.0 + 40.0;
    return 1.1 * merged_us < old_us;
}

// tuning = 0 (src/optimize.cpp:178-191): XtX_l = |l| CC' + lambda I and Xty_l = (S C')[l] - CC' sum_{r in l} s_r are the merged
// update's equations with EMPTY held-out sums, and sum_{r in l} s_r comes from the level-pair sample counts: one launch per
// covariate (k_level_merged on an all-zero record) instead of five over the samples, and R is rebuilt once per outer iteration.
// What it buys is launch latency: on small data a long unmasked fit (the reference's fit() default) is a chain of ~5 us kernels.
bool unmasked_fused(const insider_hip_handle *h, int masked)
{
    return !masked && h->merged && h->row_merged && h->row_fused && h->m == 0 && h->lvl_zero;
}

// V = C A' for the stacked levels [q_begin, q_end) (all of them once per outer iteration, then the updated covariate's)
int launch_gene_v(insider_hip_handle *h, int q_begin, int q_end)
{
    // V[:, q_begin:q_end) = C A[q_begin:q_end, :]'  (A given "transposed": one row per output column)
    const int N = q_end - q_begin;
    if (N <= 0) return INSIDER_OK;
    if (h->mm_fast && h->p >= MM_FAST_MIN) {   // A' staged in LDS once per block, C read in 16-byte pieces (k_mm_rows2)
        const int tiles = cdiv((int)h->p, 16), tpw = mm_tiles_per_wave(h, tiles);
        const size_t ldsb = (size_t)4 * cdiv(h->K, 16) * 64 * sizeof(double);
#define GV2_LAUNCH(NT_)                                                                                                       \
    hipLaunchKernelGGL((k_mm_rows2<NT_, true>), dim3(cdiv(cdiv(tiles, tpw), 4), cdiv(N, 16 * NT_)), dim3(256), ldsb * NT_, h->stream, \
                       (const double *)h->C, (int64_t)h->KP, (int)h->p, h->K,                                                \
                       (const double *)(h->Astack + (size_t)q_begin * h->KP), h->KP, N, h->Vlev + q_begin, (int64_t)h->SLP, N, tpw, h->KP, 0)
        if (N <= 16) GV2_LAUNCH(1);
        else if (N <= 32) GV2_LAUNCH(2);
        else GV2_LAUNCH(4);
#undef GV2_LAUNCH
        KCHECK();
        return INSIDER_OK;
    }
#define GV_LAUNCH(NT_)                                                                                                        \
    hipLaunchKernelGGL((k_mm_rows<NT_, true>), dim3(cdiv(cdiv((int)h->p, 16), 4), cdiv(N, 16 * NT_)), dim3(256), 0, h->stream, \
                       (const double *)h->C, (int64_t)h->KP, (int)h->p, h->K,                                                \
                       (const double *)(h->Astack + (size_t)q_begin * h->KP), h->KP, N, h->Vlev + q_begin, (int64_t)h->SLP, N)
    if (N <= 16) GV_LAUNCH(1);        // (a covariate with few levels: one 16-column tile, not four)
    else if (N <= 32) GV_LAUNCH(2);
    else GV_LAUNCH(4);
#undef GV_LAUNCH
    KCHECK();
    return INSIDER_OK;
}

int launch_row_stats(insider_hip_handle *h, bool timed)
{
    Timer t;
    int rc = t.begin(h, timed);
    if (rc) return rc;
    rc = launch_list_stats(h, false, h->nseg, h->C, h->stat);
    if (rc) return rc;
    return t.end(h, h->ev_row);
}

int do_allreduce(insider_hip_handle *h, double *buf, int64_t count)
{
    if (h->world <= 1 && !h->force_allreduce) return INSIDER_OK;
    if (h->comm) {
        // in-library RCCL: the collective is enqueued on the library's own stream, between the kernels that produce
        // and consume `buf`; no host synchronisation, no callback into the host language
        const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, h->comm, h->stream);
        if (r != ncclSuccess) return fail(INSIDER_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        return INSIDER_OK;
    }
    if (!h->allreduce) {
        if (h->world > 1) return fail(INSIDER_ERR_COMM, "world > 1 needs insider_hip_comm_init() or an all-reduce callback");
        return INSIDER_OK;
    }
    // stream-ordered: the callback enqueues the collective against h->stream (see include/insider_hip.h)
    if (h->allreduce(h->allreduce_user, buf, count, (void *)h->stream) != 0)
        return fail(INSIDER_ERR_COMM, "all-reduce callback failed");
    return INSIDER_OK;
}

// level records' Gram part of covariate i: sum_j n_jl c_j c_j' for every level -> rec[l][0 .. STAT)
int launch_level_gram(insider_hip_handle *h, int i, hipStream_t st, double *rec, const CovTables *cont_ct = nullptr)
{
    const CovTables &ct = cont_ct ? *cont_ct : h->cov[i];
    const WgPlan w = cont_ct ? WgPlan() : wgemm_plan(h, i, h->K);
    if (w.use) {
        const int stat_len = h->NB * (h->NB + 1) / 2 * 256, plen = stat_len + 2 * h->KP + 2;
        const float *hn = h->cf_hn + h->cf.hn_off[h->cf_pos[i]];
        const dim3 grid(cdiv(cdiv(w.ntile, 2) * w.nslab, 4), 1, w.zch);   // (slab, pair-tile pair) items, four waves per block
#define WG_LAUNCH(LT_)                                                                                                     \
    hipLaunchKernelGGL((k_wgemm<LT_>), grid, dim3(256), 0, st, hn, h->cf.hn_stride, w.tiles, (const double *)h->C, h->KP,   \
                       (int)h->p, w.slab, w.nslab, (const uint8_t *)h->wg_pair, w.ntile, h->wg_part)
        switch (w.LT) {
            case 4: WG_LAUNCH(4); break;
            case 5: WG_LAUNCH(5); break;
            case 6: WG_LAUNCH(6); break;
            default: WG_LAUNCH(7); break;
        }
#undef WG_LAUNCH
        KCHECK();
        hipLaunchKernelGGL(k_wgemm_sum, dim3(cdiv(stat_len, 256), ct.L), dim3(256), 0, st, (const double *)h->wg_part, w.nslab,
                           w.tiles, w.ntile, h->K, stat_len, rec, plen);
        KCHECK();
        return INSIDER_OK;
    }
    NB_DISPATCH(h->NB, {
        constexpr int STAT_ = Geo<NB_>::STAT, PLEN = STAT_ + 2 * Geo<NB_>::KP + 2;
        if (ct.nitems > 0)   // no held-out entry at all: every level sum is zero
            hipLaunchKernelGGL((k_wsyrk<NB_, WPB_>), dim3(cdiv(ct.nitems, WPB_)), dim3(WPB_ * 64), 0, st,
                               (const uint32_t *)ct.item_begin, (const uint32_t *)ct.item_end, ct.nitems,
                               (const int *)ct.wl_idx, (const double *)ct.wl_w, (const double *)h->C, (int64_t)h->p, h->wpart);
        hipLaunchKernelGGL(k_level_sum, dim3(cdiv(STAT_, 16), ct.L), dim3(256), 0, st, (const double *)h->wpart,
                           (const int *)ct.lvl_item_ptr, STAT_, rec, PLEN);
    });
    KCHECK();
    return INSIDER_OK;
}

// merged row update: the weighted SYRK + level sums of every covariate (they depend on C and the static lists only) on the
// second side stream, from the point where C is final; row_update() waits for its covariate's event
int launch_wsyrk_side(insider_hip_handle *h)
{
    HIPCHECK(hipEventRecord(h->ev_c_ready, h->stream));
    HIPCHECK(hipStreamWaitEvent(h->side2, h->ev_c_ready, 0));
    // The weighted SYRK of the first covariate is the longest kernel of the row phase (MFMA-bound on its (level, gene)
    // pairs) and the first thing the main chain waits for: it starts at once.  C'C and (S^train C') (launch_row_prep) run on
    // a third stream: they are first read by k_level_reduce, which waits for ev_prep.
    // Who is dispatched first decides the phase's length (round 4): when the main chain's k_gene_u (12500 blocks) reaches the
    // CUs before the level Gram GEMM of covariate 0 (255 blocks, one wave per SIMD), the GEMM's blocks land unevenly between
    // them and it takes 220 us instead of 140 — 590 against 460 us for the phase, the mode chosen by how the queues happen to
    // wake up after the solve.  So k_gene_u waits for this event, recorded on the GEMM's stream directly in front of it: its
    // queue goes on to dispatch the GEMM at once, the main stream's wake-up comes a few microseconds later (and behind V).
    HIPCHECK(hipEventRecord(h->ev_head, h->side2));
    HIPCHECK(hipStreamWaitEvent(h->side3, h->ev_c_ready, 0));
    if (int rp = launch_gram(h, h->C, h->p, h->CCt, h->side3, h->gram_part2)) return rp;
    if (int rp = launch_mm_reduce_kp(h, h->Strain, h->SLP, h->C, (int)h->p, h->SL, h->sc_part2, h->SC, h->side3)) return rp;
    HIPCHECK(hipEventRecord(h->ev_prep, h->side3));
    for (int i = 0; i < h->c; ++i) {
        const int plen = h->NB * (h->NB + 1) / 2 * 256 + 2 * h->KP + 2;
        if (int rg = launch_level_gram(h, i, h->side2, h->lvl_sum_all + (size_t)h->lvl_off[i] * plen)) return rg;
        // the level solves need C'C and (S^train C') as well: this stream joins the third one once, in front of its first event,
        // so that every ev_w stands for ev_prep too and the main chain waits ONCE per covariate
        if (i == 0 && (h->join_lean & 1)) HIPCHECK(hipStreamWaitEvent(h->side2, h->ev_prep, 0));
        HIPCHECK(hipEventRecord(h->ev_w[i], h->side2));
    }
    for (int k = 0; k < (h->cont_merged ? h->m : 0); ++k) {   // continuous columns: one-level covariates with real-valued weights
        const int plen = h->NB * (h->NB + 1) / 2 * 256 + 2 * h->KP + 2;
        if (int rg = launch_level_gram(h, 0, h->side2, h->lvl_sum_all + (size_t)(h->SLcat + k) * plen, &h->contm[k])) return rg;
        HIPCHECK(hipEventRecord(h->ev_w[h->c + k], h->side2));
    }
    h->w_ready = true;
    return INSIDER_OK;
}

// one covariate's row update: categorical covariate i (optimize_row, src/optimize.cpp:139-198), or continuous
// column j (optimize_continuous_v2, :76-137) when cont_col >= 0
int row_update(insider_hip_handle *h, int i, int cont_col, int masked, double lambda1, bool rebuild_R = true)
{
    const bool cont = cont_col >= 0;
    bool fused_solve = false;
    const CovTables &ct = cont ? h->cont : h->cov[i];
    const int row0 = cont ? h->SLcat + cont_col : h->lvl_off[i];   // first row of this covariate in Astack / SC
    LevelArgs la;
    la.stat = h->stat;
    la.nseg = h->nseg;
    la.n = (int)h->n;
    la.K = h->K;
    la.masked = masked;
    la.R = h->R;
    la.lev = h->lev;
    la.lvl_off = h->lvl_off_d;
    la.cov = cont ? -1 : i;
    la.own_row = row0;
    la.weights = cont ? h->Zc + (size_t)cont_col * h->n : nullptr;
    la.chunk_begin = ct.chunk_begin;
    la.chunk_end = ct.chunk_end;
    la.members = cont ? h->ident_members : h->members_all + (size_t)i * h->n;
    la.nchunks = ct.nchunks;
    la.Astack = h->Astack;
    la.part = h->lvl_part;
    LevelReduceArgs ra;
    ra.part = h->lvl_sum;
    ra.L = ct.L;
    ra.K = h->K;
    ra.CCt = h->CCt;
    ra.SC = h->SC;
    ra.sc_off = row0;
    ra.eq = h->eq;
    if (cont && use_merged(h, masked)) {
        // merged update of continuous column cont_col (optimize_continuous_v2, src/optimize.cpp:76-137): the one-level covariate
        // with real-valued membership weights z_r — u_j from the real-valued count table, Y = U'C, the level record's Gram sum
        // from the (gene, sum z^2) list, sum_r z_r s_r from the real-valued pair counts; then the reference's cyclic scalar
        // passes on (H, b) (k_cont_cd) as on the per-sample path
        const int KP = h->KP;
        const CovTables &cm = h->contm[cont_col];
        ColFacArgs ca = h->cf;
        ca.zt = h->cf_zt;
        hipLaunchKernelGGL((k_gene_uc<4>), dim3(cdiv(h->p, 4)), dim3(256), 0, h->stream, ca, cont_col, (const double *)h->Vlev, h->SLP,
                           h->U);
        KCHECK();
        int ypart_n = 0;
        if (int rcy = launch_mm_reduce_kp(h, h->U, 2, h->C, (int)h->p, 1, h->sc_part, nullptr, nullptr, &ypart_n)) return rcy;
        if (h->w_ready) {
            HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_w[h->c + cont_col], 0));
            if (!((h->join_lean & 1) && h->c > 0)) HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_prep, 0));
        }
        NB_DISPATCH(h->NB, {
            (void)WPB_;
            constexpr int STAT_ = Geo<NB_>::STAT, PLEN = STAT_ + 2 * Geo<NB_>::KP + 2;
            double *rec = h->w_ready ? h->lvl_sum_all + (size_t)row0 * PLEN : h->lvl_sum;
            if (!h->w_ready)
                if (int rg = launch_level_gram(h, 0, h->stream, rec, &cm)) return rg;
            hipLaunchKernelGGL((k_level_merged<NB_>), dim3(1), dim3(256), 0, h->stream, (const double *)rec, (const double *)h->sc_part,
                               ypart_n, (const double *)cm.paircnt, h->SL, (const double *)h->Astack, (const int *)h->one_count,
                               (const double *)h->CCt, (const double *)(h->SC + (size_t)row0 * KP), 1, h->K, lambda1, 0, h->eq,
                               h->Astack + (size_t)row0 * KP, h->failflag, (const double *)(h->cont_cnt + cont_col));
        });
    } else if (!cont && use_merged(h, masked)) {
        // merged update: one weighted rank-one term per (level, gene) pair, one look-up per held-out entry
        const int L = ct.L, LP = (int)round_up(L, 2), KP = h->KP;
        const size_t gu_lds = (size_t)4 * (h->SL + LP + GU_BATCH * WAVE) * sizeof(double);
        if (h->cf_pair_ok && (h->row_counts || h->m > 0) && h->c <= CF_MAXC && gu_lds <= 64 * 1024) {   // u from the dense pair counts (insider_col_factored.hpp)
            ColFacArgs ca = h->cf;
            ca.cnt = h->cf_cnt;
            ca.zt = h->m > 0 ? h->cf_zt : nullptr;   // (+ the continuous covariates' term from the real-valued counts)
            hipLaunchKernelGGL((k_gene_u_cnt<4>), dim3(cdiv(h->p, 4)), dim3(256), gu_lds, h->stream, ca, h->cf_pos[i], LP,
                               (const double *)h->Vlev, h->SLP, h->SL, h->U);
        } else {
            // (k_gene_u knows nothing of the continuous covariates' term: insider_hip_create_ex leaves cont_merged off when a
            // covariate's k_gene_u_cnt record does not fit, so this branch is never reached with m > 0)
            if (h->m > 0) return fail(INSIDER_ERR_UNSUPPORTED, "merged row update with continuous covariates needs the pair-count form");
            hipLaunchKernelGGL((k_gene_u<4>), dim3(cdiv(h->p, 4)), dim3(256), (size_t)4 * (h->SLcat + GU_TILE) * sizeof(double),
                               h->stream, (const uint32_t *)ct.grp, (const uint16_t *)ct.slev,
                               (size_t)h->col_entries + LIST_BLOCK, h->c - 1, L, LP, (const double *)h->Vlev, h->SLP, (int)h->p,
                               h->SLcat, h->U);
        }
        KCHECK();
        // Y = U'C, the same reduction over genes as (S C'); with the fused level kernel its per-slab partial sums are added up
        // there (k_sum_partials' order), which takes one launch per covariate off the main chain
        int ypart_n = 0;
        if (int rcy = launch_mm_reduce_kp(h, h->U, LP, h->C, (int)h->p, L, h->sc_part, h->row_fused ? nullptr : h->Ylvl, nullptr,
                                          &ypart_n))
            return rcy;
        if (!h->row_fused) ypart_n = 0;
        if (h->w_ready) {   // wsyrk + level sums came from side2, C'C and (S^train C') from side3
            HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_w[i], 0));
            if (!(h->join_lean & 1)) HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_prep, 0));
        }
        if (!h->w_ready)
            if (int rg = launch_level_gram(h, i, h->stream, h->lvl_sum)) return rg;
        NB_DISPATCH(h->NB, {
            constexpr int STAT_ = Geo<NB_>::STAT, PLEN = STAT_ + 2 * Geo<NB_>::KP + 2;
            double *rec = h->w_ready ? h->lvl_sum_all + (size_t)h->lvl_off[i] * PLEN : h->lvl_sum;
            // the level records' tail, the level equations and (unless the equations still have to cross ranks) the solves: one launch
            fused_solve = h->world <= 1 && !h->force_allreduce && h->row_fused && NB_ <= 2;
            if (h->row_fused)
                hipLaunchKernelGGL((k_level_merged<NB_>), dim3(L), dim3(256), 0, h->stream, (const double *)rec,
                                   (const double *)(ypart_n ? h->sc_part : h->Ylvl), ypart_n, (const double *)ct.paircnt, h->SL, (const double *)h->Astack,
                                   (const int *)(h->lvl_count_all + h->lvl_off[i]), (const double *)h->CCt,
                                   (const double *)(h->SC + (size_t)row0 * KP), L, h->K, lambda1, fused_solve ? 1 : 0, h->eq,
                                   h->Astack + (size_t)row0 * KP, h->failflag);
            else {
                hipLaunchKernelGGL(k_level_pack, dim3(L), dim3(256), 0, h->stream, (const double *)h->Ylvl,
                                   (const double *)ct.paircnt, h->SL, (const double *)h->Astack,
                                   (const int *)(h->lvl_count_all + h->lvl_off[i]), L, h->K, KP, STAT_, rec);
                ra.part = rec;
                hipLaunchKernelGGL((k_level_reduce<NB_>), dim3(L), dim3(64), 0, h->stream, ra);
            }
        });
    } else if (!cont && unmasked_fused(h, masked)) {
        const int L = ct.L, KP = h->KP;
        NB_DISPATCH(h->NB, {
            (void)WPB_;
            fused_solve = h->world <= 1 && !h->force_allreduce && NB_ <= 2;
            hipLaunchKernelGGL((k_level_merged<NB_>), dim3(L), dim3(256), 0, h->stream, (const double *)h->lvl_zero,
                               (const double *)h->lvl_zero, 0, (const double *)ct.paircnt, h->SL, (const double *)h->Astack,
                               (const int *)(h->lvl_count_all + h->lvl_off[i]), (const double *)h->CCt,
                               (const double *)(h->SC + (size_t)row0 * KP), L, h->K, lambda1, fused_solve ? 1 : 0, h->eq,
                               h->Astack + (size_t)row0 * KP, h->failflag);
        });
    } else {
        NB_DISPATCH(h->NB, {
            (void)WPB_;
            constexpr int PLEN = Geo<NB_>::STAT + 2 * Geo<NB_>::KP + 2;
            hipLaunchKernelGGL((k_level_partial<NB_>), dim3(ct.nchunks), dim3(64), 0, h->stream, la);
            hipLaunchKernelGGL(k_level_sum, dim3(cdiv(PLEN, 16), ct.L), dim3(256), 0, h->stream,
                               (const double *)h->lvl_part, (const int *)ct.lvl_chunk_ptr, PLEN, h->lvl_sum, PLEN);
            hipLaunchKernelGGL((k_level_reduce<NB_>), dim3(ct.L), dim3(64), 0, h->stream, ra);
        });
    }
    KCHECK();
    if (fused_solve) return rebuild_R ? launch_build_R(h) : INSIDER_OK;
    int rc = do_allreduce(h, h->eq, (int64_t)ct.L * (h->KP * h->KP + h->KP));
    if (rc) return rc;
    if (cont && masked) {
        NB_DISPATCH(h->NB, {
            (void)WPB_;
            hipLaunchKernelGGL((k_cont_cd<NB_>), dim3(1), dim3(64), 0, h->stream, (const double *)h->eq, h->K, lambda1,
                               h->Astack + (size_t)row0 * h->KP);
        });
    } else {
        NB_DISPATCH(h->NB, {
            (void)WPB_;
            hipLaunchKernelGGL((k_level_solve<NB_>), dim3(ct.L), dim3(64), 0, h->stream, (const double *)h->eq,
                               (const int *)(cont ? h->one_count : h->lvl_count_all + h->lvl_off[i]), ct.L, h->K, lambda1,
                               h->Astack + (size_t)row0 * h->KP, h->failflag);
        });
    }
    KCHECK();
    // the next covariate's Gauss-Seidel residual sees this update (:353-355, :347-349): the per-sample path reads it from R;
    // the merged update works from the stacked factors and the gene tables, so there only the last covariate rebuilds R
    return rebuild_R ? launch_build_R(h) : INSIDER_OK;
}

struct LossOut {
    double sum_residual, train_rmse, test_rmse, row_reg_half, col_reg_half, l1_reg, loss;
};

// evaluate() + compute_loss() (src/utils.cpp:56-102) from the per-gene statistics of the last column pass
int loss_checkpoint(insider_hip_handle *h, int tuning, double lambda1, double lambda2, double alpha, LossOut *o)
{
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, h->stream, h->sse_train, h->sse_test, h->b2, h->b1,
                       (int)h->p, h->Astack, h->SL, h->K, h->KP, h->loss_buf);
    KCHECK();
    // layout: [0]=sse_train [1]=sse_test [2]=sum c^2 [3]=sum |c| [4]=cnt_train [5]=cnt_test | [6]=sum a^2 (replicated)
    double cnt[2] = {tuning == 1 ? h->cnt_train : (double)h->n * (double)h->p, h->cnt_test};
    HIPCHECK(hipMemcpyAsync(h->loss_buf + 4, cnt, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = do_allreduce(h, h->loss_buf, 6);
    if (rc) return rc;
    double v[8];
    HIPCHECK(hipMemcpyAsync(v, h->loss_buf, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    o->sum_residual = v[0];
    o->train_rmse = std::sqrt(v[0] / v[4]);                                                   // :63,66
    o->test_rmse = (tuning == 1 && v[5] > 0) ? std::sqrt(v[1] / v[5]) : std::numeric_limits<double>::quiet_NaN();
    const double nfA = std::sqrt(v[6]), nfC = std::sqrt(v[2]);
    o->row_reg_half = lambda1 * nfA * nfA / 2;                                                // :83-86 (all A_i share lambda1)
    o->col_reg_half = lambda2 * (1 - alpha) * nfC * nfC / 2;                                  // :88
    o->l1_reg = lambda2 * alpha * v[3];                                                       // :91
    o->loss = o->sum_residual / 2 + o->row_reg_half + o->col_reg_half + o->l1_reg;            // :93
    return INSIDER_OK;
}

int check_fail_flag(insider_hip_handle *h)
{
    int f = 0;
    HIPCHECK(hipMemcpyAsync(&f, h->failflag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (f) {
        HIPCHECK(hipMemsetAsync(h->failflag, 0, sizeof(int), h->stream));
        return fail(INSIDER_ERR_SOLVE, "a ridge normal-equation system was not positive definite");
    }
    return INSIDER_OK;
}

// cap-hit counter and longest solve of the column updates since the last reset (failflag[2..3])
int read_cap_hits(insider_hip_handle *h)
{
    int v[2] = {0, 0};
    HIPCHECK(hipMemcpyAsync(v, h->failflag + 2, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    h->cap_hits = v[0];
    h->max_gene_sweeps = v[1];
    return INSIDER_OK;
}

void clear_events(insider_hip_handle *h)
{
    for (auto *v : {&h->ev_col, &h->ev_row, &h->ev_cd, &h->ev_test}) {
        for (auto e : *v) (void)hipEventDestroy(e);
        v->clear();
    }
}

// host row/column factors -> padded device layout (the reference aliases R's memory, src/optimize.cpp:283-284)
int upload_factors(insider_hip_handle *h, double *const *A, const double *C, int K)
{
    const int KP = h->KP;
    for (int i = 0; i < h->c; ++i) {
        const int L = h->n_levels[i];
        HIPCHECK(hipMemcpyAsync(h->stage, A[i], (size_t)L * K * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_pack_A, dim3(cdiv(L * KP, 256)), dim3(256), 0, h->stream, (const double *)h->stage, L, K, KP,
                           h->Astack + (size_t)h->lvl_off[i] * KP);
        KCHECK();
        HIPCHECK(hipStreamSynchronize(h->stream));   // stage is reused
    }
    if (h->m > 0) {
        HIPCHECK(hipMemcpyAsync(h->stage, A[h->c], (size_t)h->m * K * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_pack_A, dim3(cdiv(h->m * KP, 256)), dim3(256), 0, h->stream, (const double *)h->stage, h->m, K,
                           KP, h->Astack + (size_t)h->SLcat * KP);
        KCHECK();
        HIPCHECK(hipStreamSynchronize(h->stream));
    }
    HIPCHECK(hipMemcpyAsync(h->stage, C, (size_t)h->p * K * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_pack_rows, dim3(cdiv(h->p * KP, 256)), dim3(256), 0, h->stream, (const double *)h->stage, h->p, K,
                       KP, h->C);
    KCHECK();
    return INSIDER_OK;
}

// device factors -> the caller's buffers (the reference returns copies AND has mutated its inputs in place, :413-421)
int download_factors(insider_hip_handle *h, double *const *A, double *C, int K)
{
    const int KP = h->KP;
    if (A) {
        for (int i = 0; i < h->c; ++i) {
            const int L = h->n_levels[i];
            hipLaunchKernelGGL(k_unpack_A, dim3(cdiv(L * K, 256)), dim3(256), 0, h->stream,
                               (const double *)(h->Astack + (size_t)h->lvl_off[i] * KP), L, K, KP, h->stage);
            KCHECK();
            HIPCHECK(hipMemcpyAsync(A[i], h->stage, (size_t)L * K * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
        }
        if (h->m > 0) {
            hipLaunchKernelGGL(k_unpack_A, dim3(cdiv(h->m * K, 256)), dim3(256), 0, h->stream,
                               (const double *)(h->Astack + (size_t)h->SLcat * KP), h->m, K, KP, h->stage);
            KCHECK();
            HIPCHECK(hipMemcpyAsync(A[h->c], h->stage, (size_t)h->m * K * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
        }
    }
    if (C) {
        hipLaunchKernelGGL(k_unpack_rows, dim3(cdiv(h->p * K, 256)), dim3(256), 0, h->stream, (const double *)h->C, h->p, K,
                           KP, h->stage);
        KCHECK();
        HIPCHECK(hipMemcpyAsync(C, h->stage, (size_t)h->p * K * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
    }
    return INSIDER_OK;
}

// C C' and S_i C' for the row step (src/optimize.cpp:332 and the unmasked part of :166,188)
int launch_row_prep(insider_hip_handle *h, int masked)
{
    int rc = launch_gram(h, h->C, h->p, h->CCt);
    if (rc) return rc;
    // the merged update wants (S^train C'): the per-level sums of the TRAIN entries, i.e. (S C') minus sum_r bc_r
    if ((rc = launch_mm_reduce_kp(h, use_merged(h, masked) ? h->Strain : h->S, h->SLP, h->C, (int)h->p, h->SL, h->sc_part, h->SC)))
        return rc;
    return INSIDER_OK;
}

int check_factor_args(insider_hip_handle *h, double *const *A, const double *C, int inc_continuous, int tuning)
{
    if (!h || !A || !C) return fail(INSIDER_ERR_ARG, "null argument");
    if (tuning != 0 && tuning != 1)   // the reference prints and exit(1)s here (src/optimize.cpp:249-251)
        return fail(INSIDER_ERR_ARG, "Parameter tuning should be either 0 or 1!");
    if (inc_continuous != 0 && inc_continuous != 1)   // src/optimize.cpp:270-272
        return fail(INSIDER_ERR_ARG, "The value of prarameter inc_continuous can only be 0 or 1.");
    if (inc_continuous == 1 && h->m == 0)
        return fail(INSIDER_ERR_ARG, "inc_continuous = 1 needs a handle created with ctns_confounder (insider_hip_create_ex)");
    if (inc_continuous == 0 && h->m > 0)
        return fail(INSIDER_ERR_ARG, "this handle carries continuous covariates: pass inc_continuous = 1");
    for (int i = 0; i < h->c + (h->m > 0 ? 1 : 0); ++i) if (!A[i]) return fail(INSIDER_ERR_ARG, "null row factor");
    return INSIDER_OK;
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

#ifndef INSIDER_SOURCE_SHA
#define INSIDER_SOURCE_SHA "unknown-source-sha"   /* __graft_entry__.build() passes the hash of csrc/ + include/ (insider_amd/_build.py) */
#endif
const char *insider_hip_version(void) { return "insider_hip 0.4.0 (gfx950) src:" INSIDER_SOURCE_SHA; }

const char *insider_hip_last_error(void) { return g_err.c_str(); }

int insider_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace {
// streams and events of one handle (each handle, clones included, has its own)
hipError_t make_streams(insider_hip_handle *h)
{
    hipError_t e;
#define MS(call) do { if ((e = (call)) != hipSuccess) return e; } while (0)
    MS(hipStreamCreate(&h->stream));
    MS(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    MS(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
    MS(hipStreamCreateWithFlags(&h->side3, hipStreamNonBlocking));
    MS(hipStreamCreateWithFlags(&h->lng, hipStreamNonBlocking));
    MS(hipEventCreateWithFlags(&h->ev_long_go, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_long_done, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_prep, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_c_ready, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_head, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_a_ready, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_qfull, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_qheld, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_q_early, EV_SYNC));
    h->ev_w.assign((h->c > 0 ? h->c : 1) + h->m, nullptr);
    for (auto &ev : h->ev_w) MS(hipEventCreateWithFlags(&ev, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_cd_done, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_side_done, EV_SYNC));
    MS(hipEventCreateWithFlags(&h->ev_tab, EV_SYNC));
#undef MS
    return hipSuccess;
}

void destroy_streams(insider_hip_handle *h)
{
    for (hipStream_t *st : {&h->side, &h->side2, &h->side3, &h->lng}) { if (*st) (void)hipStreamDestroy(*st); *st = nullptr; }
    for (hipEvent_t *ev : {&h->ev_long_go, &h->ev_long_done, &h->ev_prep, &h->ev_c_ready, &h->ev_head, &h->ev_a_ready, &h->ev_qfull, &h->ev_qheld, &h->ev_cd_done,
                           &h->ev_side_done, &h->ev_tab, &h->ev_q_early}) { if (*ev) (void)hipEventDestroy(*ev); *ev = nullptr; }
    for (auto ev : h->ev_w) if (ev) (void)hipEventDestroy(ev);
    h->ev_w.clear();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = nullptr;
}

// the device arrays of the data set (everything insider_hip_create_ex builds that does not depend on K or on the factors)
void free_data_set(insider_hip_handle *h)
{
    void *ptrs[] = {h->X, h->Xt, h->codes, h->codes_t, h->lev, h->lvl_off_d, h->members_all, h->lvl_ptr_all,
                    h->lvl_count_all, h->S, h->yy_train, h->yy_all, h->col_ptr, h->row_ptr, h->col_idx, h->row_idx,
                    h->col_val, h->row_val, h->col_flag, h->Zc, h->one_count, h->ident_members, h->cont.chunk_begin, h->cont.chunk_end,
                    h->cont.lvl_chunk_ptr};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    for (auto &ct : h->cov) {
        if (ct.chunk_level) (void)hipFree(ct.chunk_level);
        if (ct.chunk_begin) (void)hipFree(ct.chunk_begin);
        if (ct.chunk_end) (void)hipFree(ct.chunk_end);
        if (ct.lvl_chunk_ptr) (void)hipFree(ct.lvl_chunk_ptr);
        for (void *q : {(void *)ct.grp, (void *)ct.slev, (void *)ct.item_begin, (void *)ct.item_end, (void *)ct.lvl_item_ptr,
                        (void *)ct.wl_idx, (void *)ct.wl_w, (void *)ct.paircnt})
            if (q) (void)hipFree(q);
    }
    if (h->Strain) (void)hipFree(h->Strain);
    if (h->Sheld) (void)hipFree(h->Sheld);
    if (h->cf_cnt) (void)hipFree(h->cf_cnt);
    if (h->cf_hn) (void)hipFree(h->cf_hn);
    if (h->cf_zt) (void)hipFree(h->cf_zt);
    if (h->cont_cnt) (void)hipFree(h->cont_cnt);
    for (size_t k = 0; k < h->contm.size(); ++k) {
        CovTables &ct = h->contm[k];
        for (void *q : {(void *)(k == 0 ? ct.wl_idx : nullptr), (void *)ct.wl_w, (void *)(k == 0 ? ct.item_begin : nullptr),
                        (void *)(k == 0 ? ct.item_end : nullptr), (void *)(k == 0 ? ct.lvl_item_ptr : nullptr), (void *)ct.paircnt})
            if (q) (void)hipFree(q);   // (the index list and the work items are shared by the columns: freed with column 0)
    }
}
}  // namespace

void insider_hip_destroy(insider_hip_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipStream_t st : {h->side, h->side2, h->side3, h->lng}) if (st) (void)hipStreamSynchronize(st);
    if (h->comm) { (void)ncclCommDestroy(h->comm); h->comm = nullptr; }
    clear_events(h);
    free_workspace(h);
    // the data set goes with its last user (insider_hip_clone shares it)
    if (!h->data_refs || h->data_refs->fetch_sub(1) == 1) {
        free_data_set(h);
        delete h->data_refs;
    }
    destroy_streams(h);
    delete h;
}

int insider_hip_clone(insider_hip_handle *src, insider_hip_handle **out)
{
    if (!out) return fail(INSIDER_ERR_ARG, "out is null");
    *out = nullptr;
    if (!src || !src->data_refs) return fail(INSIDER_ERR_ARG, "null handle");
    HIPCHECK(hipSetDevice(src->device));
    insider_hip_handle *h = new insider_hip_handle(*src);   // every data-set field and option; the rest is reset below
    forget_workspace(h);                                    // (the copied pointers are the source's buffers)
    h->stream = h->side = h->side2 = h->side3 = h->lng = nullptr;
    h->ev_long_go = h->ev_long_done = h->ev_prep = h->ev_c_ready = h->ev_head = h->ev_a_ready = h->ev_qfull = h->ev_qheld = nullptr;
    h->ev_cd_done = h->ev_side_done = h->ev_tab = h->ev_q_early = nullptr;
    h->q_kb = 0;
    h->ev_w.clear();
    for (auto *v : {&h->ev_col, &h->ev_row, &h->ev_cd, &h->ev_test}) v->clear();
    h->side_pending = h->w_ready = h->qfull_pending = h->qheld_pending = h->long_pending = false;
    h->q_kb = 0;
    h->comm = nullptr;                                      // a sharded clone joins its own communicator (insider_hip_comm_init)
    for (double &v : h->prof) v = 0.0;
    h->steady_cd_ms = h->steady_col_ms = 0.0;
    h->cap_hits = h->max_gene_sweeps = 0;
    src->data_refs->fetch_add(1);
    const hipError_t e = make_streams(h);
    if (e != hipSuccess) {
        insider_hip_destroy(h);
        return fail(e == hipErrorOutOfMemory ? INSIDER_ERR_ALLOC : INSIDER_ERR_HIP, std::string("insider_hip_clone: ") + hipGetErrorString(e));
    }
    *out = h;
    return INSIDER_OK;
}

int insider_hip_create(const double *X, int64_t n, int64_t p, const int32_t *levels, int c, const int32_t *n_levels,
                       const uint8_t *M_train, const uint8_t *M_test, int device, insider_hip_handle **out)
{
    return insider_hip_create_ex(X, n, p, levels, c, n_levels, nullptr, 0, M_train, M_test, device, out);
}

int insider_hip_create_ex(const double *X, int64_t n, int64_t p, const int32_t *levels, int c, const int32_t *n_levels,
                          const double *ctns, int m, const uint8_t *M_train, const uint8_t *M_test, int device,
                          insider_hip_handle **out)
{
    if (m < 0 || (m > 0 && !ctns)) { if (out) *out = nullptr; return fail(INSIDER_ERR_ARG, "bad continuous covariates"); }
    if (!out) return fail(INSIDER_ERR_ARG, "out is null");
    *out = nullptr;
    if (!X || !levels || !n_levels || !M_train || !M_test) return fail(INSIDER_ERR_ARG, "null input");
    if (n < 1 || p < 1 || c < 1) return fail(INSIDER_ERR_ARG, "n, p, c must be positive");
    if (n > (1 << 30) || p > (1 << 30)) return fail(INSIDER_ERR_ARG, "dimension too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(INSIDER_ERR_NO_DEVICE, "no HIP device visible: libinsider_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(INSIDER_ERR_ARG, "bad device ordinal");
    // level ids must be exactly 1..L_i (src/optimize.cpp:175,286)
    for (int i = 0; i < c; ++i) {
        if (n_levels[i] < 1) return fail(INSIDER_ERR_ARG, "n_levels must be positive");
        for (int64_t r = 0; r < n; ++r) {
            const int32_t l = levels[r + (size_t)i * n];
            if (l < 1 || l > n_levels[i]) return fail(INSIDER_ERR_ARG, "level ids must be within 1..L_i");
        }
    }
    HIPCHECK(hipSetDevice(device));
    insider_hip_handle *h = new insider_hip_handle();
    h->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_simd = 4 * cus;
    }
    h->n = n;
    h->p = p;
    h->c = c;
    h->ldn = round_up(n, CHUNK);
    h->ldp = round_up(p, CHUNK);
    h->n_levels.assign(n_levels, n_levels + c);
    h->lvl_off.assign(c + 1, 0);
    for (int i = 0; i < c; ++i) h->lvl_off[i + 1] = h->lvl_off[i] + n_levels[i];
    h->m = m;
    h->SLcat = h->lvl_off[c];
    h->SL = h->SLcat + m;
    h->SLP = (int)round_up(h->SL, 2);
    int rc = INSIDER_OK;
#define CR(x) do { rc = (x); if (rc) { insider_hip_destroy(h); return rc; } } while (0)
#define CH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { insider_hip_destroy(h); \
        return fail(e_ == hipErrorOutOfMemory ? INSIDER_ERR_ALLOC : INSIDER_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
    h->data_refs = new std::atomic<int>(1);
    CH(make_streams(h));
    // ---- X (gene-major lines of pitch ldn) and mask codes -------------------------------------------------
    CR(dmalloc(&h->X, (size_t)p * h->ldn));
    CR(dmalloc(&h->codes, (size_t)p * h->ldn));
    CH(hipMemsetAsync(h->X, 0, (size_t)p * h->ldn * sizeof(double), h->stream));
    CH(hipMemcpy2DAsync(h->X, h->ldn * sizeof(double), X, n * sizeof(double), n * sizeof(double), p,
                        hipMemcpyHostToDevice, h->stream));
    {
        uint8_t *mtr = nullptr, *mte = nullptr;
        CR(dmalloc(&mtr, (size_t)n * p));
        CR(dmalloc(&mte, (size_t)n * p));
        CH(hipMemcpyAsync(mtr, M_train, (size_t)n * p, hipMemcpyHostToDevice, h->stream));
        CH(hipMemcpyAsync(mte, M_test, (size_t)n * p, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_make_codes, dim3(cdiv(p * h->ldn, 256)), dim3(256), 0, h->stream, mtr, mte, n, p, h->ldn,
                           h->codes);
        CH(hipGetLastError());
        CH(hipStreamSynchronize(h->stream));
        (void)hipFree(mtr);
        (void)hipFree(mte);
    }
    // ---- transposed copies for the row-side pass (sample-major lines of pitch ldp) --------------------------
    CR(dmalloc(&h->Xt, (size_t)n * h->ldp));
    CR(dmalloc(&h->codes_t, (size_t)n * h->ldp));
    CH(hipMemsetAsync(h->Xt, 0, (size_t)n * h->ldp * sizeof(double), h->stream));
    CH(hipMemsetAsync(h->codes_t, CODE_TRAIN, (size_t)n * h->ldp, h->stream));
    {
        dim3 grid(cdiv(n, 32), cdiv(p, 32));   // input: p lines (rows) x n columns
        hipLaunchKernelGGL((k_transpose<double>), grid, dim3(256), 0, h->stream, (const double *)h->X, p, n, h->ldn,
                           h->Xt, h->ldp);
        hipLaunchKernelGGL((k_transpose<uint8_t>), grid, dim3(256), 0, h->stream, (const uint8_t *)h->codes, p, n,
                           h->ldn, h->codes_t, h->ldp);
        CH(hipGetLastError());
    }
    // ---- level tables ------------------------------------------------------------------------------------------
    {
        std::vector<int> lev0((size_t)c * n), members((size_t)c * n), lvl_ptr((size_t)h->SLcat + c), lvl_count(h->SLcat);
        h->cov.resize(c);
        for (int i = 0; i < c; ++i) {
            const int L = n_levels[i];
            std::vector<int> cnt(L, 0);
            for (int64_t r = 0; r < n; ++r) {
                const int l = levels[r + (size_t)i * n] - 1;
                lev0[(size_t)i * n + r] = l;
                cnt[l]++;
            }
            int *ptr = lvl_ptr.data() + h->lvl_off[i] + i;
            ptr[0] = 0;
            for (int l = 0; l < L; ++l) { ptr[l + 1] = ptr[l] + cnt[l]; lvl_count[h->lvl_off[i] + l] = cnt[l]; }
            std::vector<int> fill(ptr, ptr + L);
            for (int64_t r = 0; r < n; ++r) members[(size_t)i * n + fill[lev0[(size_t)i * n + r]]++] = (int)r;
            // chunk tables for the two-stage level reduction
            std::vector<int> ch_level, ch_begin, ch_end, lcp(L + 1, 0);
            for (int l = 0; l < L; ++l) {
                lcp[l] = (int)ch_level.size();
                for (int b = ptr[l]; b < ptr[l + 1]; b += LEVEL_CHUNK) {
                    ch_level.push_back(l);
                    ch_begin.push_back(b);
                    ch_end.push_back(std::min(b + LEVEL_CHUNK, ptr[l + 1]));
                }
            }
            lcp[L] = (int)ch_level.size();
            CovTables &ct = h->cov[i];
            ct.L = L;
            ct.nchunks = (int)ch_level.size();
            h->max_chunks = std::max(h->max_chunks, ct.nchunks);
            h->max_L = std::max(h->max_L, L);
            CR(dmalloc(&ct.chunk_level, ch_level.size()));
            CR(dmalloc(&ct.chunk_begin, ch_begin.size()));
            CR(dmalloc(&ct.chunk_end, ch_end.size()));
            CR(dmalloc(&ct.lvl_chunk_ptr, lcp.size()));
            CH(hipMemcpy(ct.chunk_level, ch_level.data(), ch_level.size() * sizeof(int), hipMemcpyHostToDevice));
            CH(hipMemcpy(ct.chunk_begin, ch_begin.data(), ch_begin.size() * sizeof(int), hipMemcpyHostToDevice));
            CH(hipMemcpy(ct.chunk_end, ch_end.data(), ch_end.size() * sizeof(int), hipMemcpyHostToDevice));
            CH(hipMemcpy(ct.lvl_chunk_ptr, lcp.data(), lcp.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        CR(dmalloc(&h->lev, lev0.size()));
        CR(dmalloc(&h->members_all, members.size()));
        CR(dmalloc(&h->lvl_ptr_all, lvl_ptr.size()));
        CR(dmalloc(&h->lvl_count_all, lvl_count.size()));
        CR(dmalloc(&h->lvl_off_d, h->lvl_off.size()));
        CH(hipMemcpy(h->lev, lev0.data(), lev0.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->members_all, members.data(), members.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->lvl_ptr_all, lvl_ptr.data(), lvl_ptr.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->lvl_count_all, lvl_count.data(), lvl_count.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->lvl_off_d, h->lvl_off.data(), h->lvl_off.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // ---- continuous covariates: one pseudo-level whose members are all samples, weighted by z ----------------------------
    if (m > 0) {
        CR(dmalloc(&h->Zc, (size_t)m * n));
        CH(hipMemcpy(h->Zc, ctns, (size_t)m * n * sizeof(double), hipMemcpyHostToDevice));   // n x m column-major == m x n rows
        std::vector<int> ident(n), cb, ce, lcp(2, 0);
        for (int64_t r = 0; r < n; ++r) ident[r] = (int)r;
        for (int64_t b0 = 0; b0 < n; b0 += LEVEL_CHUNK) { cb.push_back((int)b0); ce.push_back((int)std::min<int64_t>(b0 + LEVEL_CHUNK, n)); }
        lcp[1] = (int)cb.size();
        h->cont.L = 1;
        h->cont.nchunks = (int)cb.size();
        h->max_chunks = std::max(h->max_chunks, h->cont.nchunks);
        h->max_L = std::max(h->max_L, 1);
        CR(dmalloc(&h->ident_members, ident.size()));
        CR(dmalloc(&h->cont.chunk_begin, cb.size()));
        CR(dmalloc(&h->cont.chunk_end, ce.size()));
        CR(dmalloc(&h->cont.lvl_chunk_ptr, lcp.size()));
        CR(dmalloc(&h->one_count, 1));
        const int one = 1;
        CH(hipMemcpy(h->ident_members, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->cont.chunk_begin, cb.data(), cb.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->cont.chunk_end, ce.data(), ce.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->cont.lvl_chunk_ptr, lcp.data(), lcp.size() * sizeof(int), hipMemcpyHostToDevice));
        CH(hipMemcpy(h->one_count, &one, sizeof(int), hipMemcpyHostToDevice));
    }
    // ---- factor-independent statistics -----------------------------------------------------------------------------
    CR(dmalloc(&h->S, (size_t)p * h->SLP));
    CR(dmalloc(&h->yy_train, (size_t)p));
    CR(dmalloc(&h->yy_all, (size_t)p));
    CH(hipMemsetAsync(h->S, 0, (size_t)p * h->SLP * sizeof(double), h->stream));
    {
        unsigned long long *cnt = nullptr;
        CR(dmalloc(&cnt, 2));
        CH(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), h->stream));
        hipLaunchKernelGGL(k_line_sumsq, dim3(cdiv(p, 4)), dim3(256), 0, h->stream, (const double *)h->X,
                           (const uint8_t *)h->codes, h->ldn, (int)n, (int)p, h->yy_train, h->yy_all, cnt);
        hipLaunchKernelGGL(k_level_sums, dim3(cdiv(p * h->SLcat, 256)), dim3(256), 0, h->stream, (const double *)h->X,
                           (const uint8_t *)nullptr, h->ldn, (int)p, (const int *)h->members_all,
                           (const int *)h->lvl_ptr_all, (const int *)h->lvl_off_d, c, (int)n, h->SLcat, h->SLP, h->S);
        {   // train-only sums for the merged masked row update / the factored column statistics (the categorical columns)
            CR(dmalloc(&h->Strain, (size_t)p * h->SLP));
            CH(hipMemsetAsync(h->Strain, 0, (size_t)p * h->SLP * sizeof(double), h->stream));
            hipLaunchKernelGGL(k_level_sums, dim3(cdiv(p * h->SLcat, 256)), dim3(256), 0, h->stream, (const double *)h->X,
                               (const uint8_t *)h->codes, h->ldn, (int)p, (const int *)h->members_all,
                               (const int *)h->lvl_ptr_all, (const int *)h->lvl_off_d, c, (int)n, h->SLcat, h->SLP,
                               h->Strain);
        }
        if (m > 0)
            hipLaunchKernelGGL(k_cont_sums, dim3(cdiv(p * m, 256)), dim3(256), 0, h->stream, (const double *)h->X, h->ldn,
                               (int)p, (const double *)h->Zc, m, (int)n, h->SLcat, h->SLP, h->S);
        CH(hipGetLastError());
        unsigned long long hc[2];
        CH(hipMemcpyAsync(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, h->stream));
        CH(hipStreamSynchronize(h->stream));
        (void)hipFree(cnt);
        h->cnt_train = (double)hc[0];
        h->cnt_test = (double)hc[1];
        h->no_na = hc[0] + hc[1] == (unsigned long long)n * (unsigned long long)p;
    }
    // ---- held-out lists of both sides (the masks never change: built once) -------------------------------------------
    if (n >= LIST_PAD || p >= LIST_PAD) { insider_hip_destroy(h); return fail(INSIDER_ERR_UNSUPPORTED, "n and p must be below 2^23"); }
    for (int side = 0; side < 2; ++side) {
        const bool cols = side == 0;
        const int lines = cols ? (int)p : (int)n, len = cols ? (int)n : (int)p;
        const double *vals = cols ? h->X : h->Xt;
        const uint8_t *cds = cols ? h->codes : h->codes_t;
        const int64_t pitch = cols ? h->ldn : h->ldp;
        int *cnt_d = nullptr;
        CR(dmalloc(&cnt_d, (size_t)lines));
        hipLaunchKernelGGL(k_count_heldout, dim3(cdiv(lines, 4)), dim3(256), 0, h->stream, cds, pitch, len, lines, cnt_d);
        CH(hipGetLastError());
        std::vector<int> cnt(lines);
        CH(hipMemcpyAsync(cnt.data(), cnt_d, (size_t)lines * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        CH(hipStreamSynchronize(h->stream));
        (void)hipFree(cnt_d);
        std::vector<uint32_t> ptr((size_t)lines + 1);
        uint64_t tot = 0;
        for (int i = 0; i < lines; ++i) { ptr[i] = (uint32_t)tot; tot += (uint64_t)round_up(cnt[i], LIST_ALIGN); }
        ptr[lines] = (uint32_t)tot;
        if (tot >= (1ull << 32)) { insider_hip_destroy(h); return fail(INSIDER_ERR_UNSUPPORTED, "more than 2^32 held-out entries"); }
        uint32_t *&dptr = cols ? h->col_ptr : h->row_ptr;
        int *&didx = cols ? h->col_idx : h->row_idx;
        double *&dval = cols ? h->col_val : h->row_val;
        (cols ? h->col_entries : h->row_entries) = tot;
        CR(dmalloc(&dptr, ptr.size()));
        CR(dmalloc(&didx, (size_t)tot + LIST_BLOCK));
        CR(dmalloc(&dval, (size_t)tot + LIST_BLOCK));
        CH(hipMemcpyAsync(dptr, ptr.data(), ptr.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
        uint8_t *dflag = nullptr;
        if (cols && !h->no_na) { CR(dmalloc(&h->col_flag, (size_t)tot + LIST_BLOCK)); dflag = h->col_flag; }
        hipLaunchKernelGGL(k_fill_lists, dim3(cdiv(lines, 4)), dim3(256), 0, h->stream, vals, cds, pitch, len, lines,
                           (const uint32_t *)dptr, didx, dval, dflag);
        CH(hipGetLastError());
        CH(hipStreamSynchronize(h->stream));
    }
    // ---- merged masked row update: per covariate, the genes' held-out samples grouped by level, the (gene, count)
    // lists of every level and the level-pair sample counts (insider_row_merged.hpp) -------------------------------------
    // (k_gene_u keeps a gene's SLcat look-up values per wave in LDS: beyond ~1500 stacked levels the per-sample path stays)
    // (with continuous covariates, m <= 4: the same tables serve the pair-count column statistics, ColFacArgs::zt)
    if (m <= 4 && (size_t)4 * (h->SLcat + GU_TILE) * sizeof(double) <= 64 * 1024) {
        constexpr uint32_t SEG = 1024;   // list entries per weighted-SYRK work item (multiple of LIST_ALIGN)
        std::vector<int> lev0((size_t)c * n);
        CH(hipMemcpy(lev0.data(), h->lev, lev0.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < c; ++i) {
            CovTables &ct = h->cov[i];
            const int L = ct.L;
            CR(dmalloc(&ct.grp, (size_t)p * (L + 1)));
            const size_t plane = (size_t)h->col_entries + LIST_BLOCK;
            if (h->SLcat > 65535) { insider_hip_destroy(h); return fail(INSIDER_ERR_UNSUPPORTED, "more than 65535 levels in total"); }
            CR(dmalloc(&ct.slev, plane * (size_t)std::max(c - 1, 1)));
            const size_t lds = (size_t)4 * L * sizeof(int);
            if (lds > 60 * 1024) { insider_hip_destroy(h); return fail(INSIDER_ERR_UNSUPPORTED, "a covariate has more than 3840 levels"); }
            hipLaunchKernelGGL(k_group_count, dim3(cdiv(p, 4)), dim3(256), lds, h->stream, (const uint32_t *)h->col_ptr,
                               (const int *)h->col_idx, (const int *)(h->lev + (size_t)i * n), L, (int)p, ct.grp);
            hipLaunchKernelGGL(k_group_fill, dim3(cdiv(p, 4)), dim3(256), lds, h->stream, (const uint32_t *)h->col_ptr,
                               (const int *)h->col_idx, (const int *)h->lev, (const int *)h->lvl_off_d, c, (int)n, i, L,
                               (int)p, (const uint32_t *)ct.grp, ct.slev, plane);
            CH(hipGetLastError());
            std::vector<uint32_t> grp((size_t)p * (L + 1));
            CH(hipMemcpyAsync(grp.data(), ct.grp, grp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            CH(hipStreamSynchronize(h->stream));
            // (gene, count) list of every level, padded to LIST_ALIGN; work items of at most SEG entries
            std::vector<int> widx, lip(L + 1, 0);
            std::vector<double> ww;
            std::vector<uint32_t> ib, ie;
            for (int l = 0; l < L; ++l) {
                lip[l] = (int)ib.size();
                const size_t start = widx.size();
                for (int64_t j = 0; j < p; ++j) {
                    const uint32_t cnt = grp[(size_t)j * (L + 1) + l + 1] - grp[(size_t)j * (L + 1) + l];
                    if (cnt) { widx.push_back((int)j); ww.push_back((double)cnt); }
                }
                while ((widx.size() - start) % LIST_ALIGN) { widx.push_back(LIST_PAD); ww.push_back(0.0); }
                for (size_t b = start; b < widx.size(); b += SEG) {
                    ib.push_back((uint32_t)b);
                    ie.push_back((uint32_t)std::min(b + SEG, widx.size()));
                }
            }
            lip[L] = (int)ib.size();
            if (widx.size() >= (1ull << 32)) { insider_hip_destroy(h); return fail(INSIDER_ERR_UNSUPPORTED, "level lists too long"); }
            ct.nitems = (int)ib.size();
            ct.npairs = 0;
            for (double wv : ww) ct.npairs += wv != 0.0;
            h->max_items = std::max(h->max_items, ct.nitems);
            CR(dmalloc(&ct.wl_idx, widx.size() + LIST_BLOCK));
            CR(dmalloc(&ct.wl_w, ww.size() + LIST_BLOCK));
            CR(dmalloc(&ct.item_begin, ib.size() + 1));
            CR(dmalloc(&ct.item_end, ie.size() + 1));
            CR(dmalloc(&ct.lvl_item_ptr, lip.size()));
            CH(hipMemcpy(ct.wl_idx, widx.data(), widx.size() * sizeof(int), hipMemcpyHostToDevice));
            CH(hipMemcpy(ct.wl_w, ww.data(), ww.size() * sizeof(double), hipMemcpyHostToDevice));
            CH(hipMemcpy(ct.item_begin, ib.data(), ib.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            CH(hipMemcpy(ct.item_end, ie.data(), ie.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            CH(hipMemcpy(ct.lvl_item_ptr, lip.data(), lip.size() * sizeof(int), hipMemcpyHostToDevice));
            // samples per (level of covariate i, stacked level of another covariate): sum_{r in l} s_r = paircnt A
            // (+ m columns sum_{r in l} z_rk: a continuous column is a stacked "level" with real-valued counts)
            std::vector<double> pc((size_t)L * h->SL, 0.0);
            for (int64_t r = 0; r < n; ++r) {
                const int l = lev0[(size_t)i * n + r];
                for (int q = 0; q < c; ++q)
                    if (q != i) pc[(size_t)l * h->SL + h->lvl_off[q] + lev0[(size_t)q * n + r]] += 1.0;
                for (int k = 0; k < m; ++k) pc[(size_t)l * h->SL + h->SLcat + k] += ctns[(size_t)k * n + r];
            }
            CR(dmalloc(&ct.paircnt, pc.size()));
            CH(hipMemcpy(ct.paircnt, pc.data(), pc.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        h->merged = true;
        // ---- factored column statistics: covariates by decreasing level count, the planes of the later ones ----------
        CR(dmalloc(&h->Sheld, (size_t)p * h->SLP));
        hipLaunchKernelGGL(k_sub, dim3(cdiv((int64_t)p * h->SLP, 256)), dim3(256), 0, h->stream, (const double *)h->S,
                           (const double *)h->Strain, (size_t)p * h->SLP, h->Sheld);
        CH(hipGetLastError());
        if (c <= CF_MAXC) {
            std::vector<int> ord(c);
            for (int i = 0; i < c; ++i) ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return n_levels[x] > n_levels[y]; });
            ColFacArgs &cf = h->cf;
            cf.p = (int)p;
            cf.c = c;
            cf.plane = (size_t)h->col_entries + LIST_BLOCK;
            for (int t = 0; t < c; ++t) {
                const int o = ord[t];
                h->cf_pos[o] = t;
                cf.grp[t] = h->cov[o].grp;
                cf.slev[t] = h->cov[o].slev;
                cf.L[t] = n_levels[o];
                cf.off[t] = h->lvl_off[o];
                cf.nlater[t] = c - 1 - t;
                for (int k = t + 1; k < c; ++k) cf.later_plane[t][k - t - 1] = ord[k] < o ? ord[k] : ord[k] - 1;
            }
            cf.tab_skip_lo = h->lvl_off[ord[0]];
            cf.tab_skip_n = n_levels[ord[0]];
            cf.tab_rows = h->SLcat - cf.tab_skip_n;
            // pair-count form: the dense per-gene count tables (one byte per cell), when they are small enough
            cf.nsteps = (cf.tab_rows + 3) / 4;
            bool fits = cf.nsteps <= CP_MAXSTEPS;
            int off = 0;
            for (int t = 0; t < c; ++t) {
                const int cells = ((cf.L[t] + 15) / 16) * 64 * (cf.nsteps <= 4 ? 4 : 8);   // bytes: 4 or 8 per lane and block
                cf.cnt_off[t] = off;
                if (cf.nlater[t] > 0) off += cells;
            }
            cf.cnt_stride = off;
            cf.cnt = nullptr;
            if (fits && off <= 64 * 1024 && (size_t)p * (size_t)off <= ((size_t)1 << 32)) {   // <= 64 KB of counts per gene
                int *ovf = nullptr;
                if (off > 0) {
                    CR(dmalloc(&h->cf_cnt, (size_t)p * off));
                    CR(dmalloc(&ovf, 1));
                    CH(hipMemsetAsync(ovf, 0, sizeof(int), h->stream));
                    hipLaunchKernelGGL((k_pair_count_build<2>), dim3(cdiv(p, 2)), dim3(128), 0, h->stream, cf, h->cf_cnt, ovf);
                    CH(hipGetLastError());
                    int hv = 0;
                    CH(hipMemcpyAsync(&hv, ovf, sizeof(int), hipMemcpyDeviceToHost, h->stream));
                    CH(hipStreamSynchronize(h->stream));
                    (void)hipFree(ovf);
                    fits = hv == 0;
                }
                if (fits && n < ((int64_t)1 << 24)) {   // (1/2 n as a float is exact)
                    int hoff = 0;
                    for (int t = 0; t < c; ++t) {
                        cf.hn_off[t] = hoff;
                        hoff += ((cf.L[t] + 15) / 16) * 16;
                    }
                    cf.hn_stride = hoff;
                    CR(dmalloc(&h->cf_hn, (size_t)(p + 4) * hoff));   // + 4 zero rows: k_wgemm reads whole steps of four genes
                    CH(hipMemsetAsync(h->cf_hn, 0, (size_t)(p + 4) * hoff * sizeof(float), h->stream));
                    hipLaunchKernelGGL(k_half_counts, dim3((unsigned)cdiv((int64_t)p * hoff, 256)), dim3(256), 0, h->stream, cf,
                                       h->cf_hn);
                    CH(hipGetLastError());
                } else {
                    fits = false;
                }
                h->cf_pair_ok = fits;
            }
            cf.zt = nullptr;
            cf.m = m;
            cf.SLcat = h->SLcat;
            for (int t = 0; t < c; ++t) cf.pos_cov[t] = ord[t];
            // The merged row update with continuous columns takes u_j from k_gene_u_cnt ALONE (only it adds the term of the
            // real-valued counts, ColFacArgs::zt; k_gene_u reads the categorical columns of V only): every covariate's launch
            // of it must fit its per-wave LDS record — V row [SL] | out [LP] | GU_BATCH x 64 partial sums, four waves per block
            // — or the whole data set stays on the per-sample / per-entry paths (a covariate with ~770 levels or more).
            bool gu_fits = true;
            for (int t = 0; t < c; ++t)
                gu_fits = gu_fits && (size_t)4 * (h->SL + round_up(h->n_levels[t], 2) + GU_BATCH * WAVE) * sizeof(double) <= 64 * 1024;
            if (m > 0 && h->cf_pair_ok && gu_fits) {
                // continuous covariates on the pair-count form: one more position (the m columns as pseudo-levels, no count
                // bytes, 1/2 n = 0: sixteen more zero floats per gene would do, the block reads what follows its offset ->
                // its own zero region) and the real-valued count table
                cf.L[c] = m;
                cf.off[c] = h->SLcat;
                cf.nlater[c] = 0;
                cf.cnt_off[c] = 0;
                int zoff = 0;
                for (int t = 0; t < c; ++t) { cf.zt_off[t] = zoff; zoff += ((cf.L[t] + 15) / 16) * 64; }
                cf.zt_off[c] = zoff;
                cf.zt_stride = zoff + 64;
                // 1/2 n of position c: a zero region behind the table (rebuilt with the longer stride)
                (void)hipFree(h->cf_hn);
                h->cf_hn = nullptr;
                cf.hn_off[c] = cf.hn_stride;
                cf.hn_stride += 16;
                CR(dmalloc(&h->cf_hn, (size_t)(p + 4) * cf.hn_stride));
                CH(hipMemsetAsync(h->cf_hn, 0, (size_t)(p + 4) * cf.hn_stride * sizeof(float), h->stream));
                hipLaunchKernelGGL(k_half_counts, dim3((unsigned)cdiv((int64_t)p * cf.hn_stride, 256)), dim3(256), 0, h->stream, cf,
                                   h->cf_hn);
                CH(hipGetLastError());
                CR(dmalloc(&h->cf_zt, (size_t)p * cf.zt_stride));
                CH(hipMemsetAsync(h->cf_zt, 0, (size_t)p * cf.zt_stride * sizeof(double), h->stream));
                // (Sheld's continuous columns: sum over the held-out entries of x z, next to S - S^train of the categorical ones)
                hipLaunchKernelGGL(k_zt_build, dim3(cdiv(p, 4)), dim3(256), 0, h->stream, cf, (const uint32_t *)h->col_ptr,
                                   (const int *)h->col_idx, (const double *)h->col_val, (const int *)h->lev, (const double *)h->Zc,
                                   (int)n, h->cf_zt, h->Sheld, h->SLP, (const double *)h->S, h->Strain);
                CH(hipGetLastError());
                CH(hipStreamSynchronize(h->stream));
                // ---- the continuous columns as one-level covariates of the merged row update ----
                h->contm.assign(m, CovTables());
                const size_t plen_l = (size_t)round_up(p, LIST_ALIGN);
                std::vector<int> widx(plen_l, LIST_PAD);
                for (int64_t j = 0; j < p; ++j) widx[j] = (int)j;
                std::vector<uint32_t> ib, ie;
                for (size_t b = 0; b < plen_l; b += SEG) { ib.push_back((uint32_t)b); ie.push_back((uint32_t)std::min(b + SEG, plen_l)); }
                const int lip[2] = {0, (int)ib.size()};
                int *d_idx = nullptr, *d_lip = nullptr;
                uint32_t *d_ib = nullptr, *d_ie = nullptr;
                // (owned by contm[0] from the moment they exist: free_data_set releases them on every error path below)
                CR(dmalloc(&d_idx, plen_l + LIST_BLOCK));
                h->contm[0].wl_idx = d_idx;
                CR(dmalloc(&d_ib, ib.size() + 1));
                h->contm[0].item_begin = d_ib;
                CR(dmalloc(&d_ie, ie.size() + 1));
                h->contm[0].item_end = d_ie;
                CR(dmalloc(&d_lip, 2));
                h->contm[0].lvl_item_ptr = d_lip;
                CH(hipMemcpy(d_idx, widx.data(), plen_l * sizeof(int), hipMemcpyHostToDevice));
                CH(hipMemcpy(d_ib, ib.data(), ib.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                CH(hipMemcpy(d_ie, ie.data(), ie.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                CH(hipMemcpy(d_lip, lip, sizeof(lip), hipMemcpyHostToDevice));
                h->max_items = std::max(h->max_items, (int)ib.size());
                std::vector<double> zz((size_t)m * m, 0.0);
                for (int k = 0; k < m; ++k)
                    for (int k2 = 0; k2 < m; ++k2) {
                        double acc = 0.0;
                        for (int64_t r = 0; r < n; ++r) acc += ctns[(size_t)k * n + r] * ctns[(size_t)k2 * n + r];
                        zz[(size_t)k * m + k2] = acc;
                    }
                std::vector<double> cc(m);
                for (int k = 0; k < m; ++k) {
                    CovTables &ct = h->contm[k];
                    ct.L = 1;
                    ct.wl_idx = d_idx;
                    ct.item_begin = d_ib;
                    ct.item_end = d_ie;
                    ct.lvl_item_ptr = d_lip;
                    ct.nitems = (int)ib.size();
                    ct.npairs = p;
                    CR(dmalloc(&ct.wl_w, plen_l + LIST_BLOCK));
                    CH(hipMemsetAsync(ct.wl_w, 0, (plen_l + LIST_BLOCK) * sizeof(double), h->stream));
                    hipLaunchKernelGGL(k_cont_weights, dim3(cdiv(p, 256)), dim3(256), 0, h->stream, (const double *)h->cf_zt, cf.zt_stride,
                                       cf.zt_off[c], k, (int)p, ct.wl_w);
                    CH(hipGetLastError());
                    std::vector<double> pc((size_t)h->SL, 0.0);
                    for (int64_t r = 0; r < n; ++r)
                        for (int q = 0; q < c; ++q) pc[h->lvl_off[q] + lev0[(size_t)q * n + r]] += ctns[(size_t)k * n + r];
                    for (int k2 = 0; k2 < m; ++k2) pc[h->SLcat + k2] = k2 == k ? 0.0 : zz[(size_t)k * m + k2];
                    CR(dmalloc(&ct.paircnt, pc.size()));
                    CH(hipMemcpy(ct.paircnt, pc.data(), pc.size() * sizeof(double), hipMemcpyHostToDevice));
                    cc[k] = zz[(size_t)k * m + k];
                }
                CR(dmalloc(&h->cont_cnt, (size_t)m));
                CH(hipMemcpy(h->cont_cnt, cc.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice));
                CH(hipStreamSynchronize(h->stream));
                h->cont_merged = true;
            }
        }
    }
    // the transposed copies were only needed to build the row-side lists
    (void)hipFree(h->Xt);
    (void)hipFree(h->codes_t);
    h->Xt = nullptr;
    h->codes_t = nullptr;
#undef CR
#undef CH
    *out = h;
    return INSIDER_OK;
}

int insider_hip_set_shard(insider_hip_handle *h, int64_t gene_offset, int rank, int world, insider_allreduce_fn fn,
                          void *user)
{
    if (!h || world < 1 || rank < 0 || rank >= world || gene_offset < 0) return fail(INSIDER_ERR_ARG, "bad shard");
    // world > 1 without a callback is completed by insider_hip_comm_init(); optimize() refuses to run with neither.
    // Installing a callback drops a communicator of an earlier insider_hip_comm_init(): the callback is then the exchange.
    if (fn && h->comm) {
        (void)hipSetDevice(h->device);
        (void)ncclCommDestroy(h->comm);
        h->comm = nullptr;
    }
    h->gene_offset = gene_offset;
    h->rank = rank;
    h->world = world;
    h->allreduce = fn;
    h->allreduce_user = user;
    return INSIDER_OK;
}

int insider_hip_comm_unique_id(void *out, int out_bytes)
{
    if (!out || out_bytes < (int)sizeof(ncclUniqueId)) return fail(INSIDER_ERR_ARG, "unique-id buffer too small (INSIDER_COMM_ID_BYTES)");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(INSIDER_ERR_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memset(out, 0, (size_t)out_bytes);
    std::memcpy(out, &id, sizeof(id));
    return INSIDER_OK;
}

int insider_hip_comm_init(insider_hip_handle *h, const void *unique_id, int rank, int world)
{
    if (!h || !unique_id || world < 1 || rank < 0 || rank >= world) return fail(INSIDER_ERR_ARG, "bad communicator arguments");
    if (h->world != world || h->rank != rank) return fail(INSIDER_ERR_ARG, "rank / world differ from insider_hip_set_shard()");
    HIPCHECK(hipSetDevice(h->device));
    if (h->comm) { (void)ncclCommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&h->comm, world, id, rank);
    if (r != ncclSuccess) { h->comm = nullptr; return fail(INSIDER_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    return INSIDER_OK;
}

int insider_hip_set_option(insider_hip_handle *h, const char *name, double value)
{
    if (!h || !name) return fail(INSIDER_ERR_ARG, "null");
    const std::string s(name);
    if (s == "max_sweeps") h->max_sweeps = value < 1 ? 1 : (int)value;
    else if (s == "order_mode") h->order_mode = (int)value;
    else if (s == "profile") h->profile = (int)value;
    else if (s == "verbose") h->verbose = (int)value;
    else if (s == "force_allreduce") h->force_allreduce = (int)value;   // call the all-reduce callback even when world == 1
    else if (s == "col_factored") h->col_factored = (int)value;   // 1 = cost model picks list / look-up / pair-count form (default), 2 = look-up form, 3 = pair-count form, 0 = k_list_stats
    else if (s == "row_counts") h->row_counts = (int)value;   // 1 = the merged row update takes u from the dense pair counts when they exist (default), 0 = from the entry lists
    else if (s == "row_merged") h->row_merged = (int)value;   // 1 = merged masked row update when the time model favours it (default), 2 = always, 0 = per-sample statistics
    else if (s == "row_gemm_waves") { h->wg_waves = std::max(64, (int)value); h->K = 0; }   // (re-plans the workspace)
    else if (s == "row_gemm") h->row_gemm = (int)value;       // 1 (default) = k_wgemm for covariates with >= 49 levels, 0 = k_wsyrk everywhere
    else if (s == "row_head") h->row_head = (int)value;       // 1 (default) = the main chain's k_gene_u is dispatched behind the level Gram GEMM (launch_wsyrk_side)
    else if (s == "row_fused") h->row_fused = (int)value;     // 1 (default) = k_level_merged (one launch per covariate), 0 = k_level_pack / k_level_reduce / k_level_solve
    else if (s == "cd_cold_iters") h->cd_cold_iters = (int)value;   // outer iterations 0 .. value-1 of a call solve in passes
    else if (s == "cd_pass1") h->cd_pass_first = (int)value;        // sweep index where the first pass stops (0 = single pass)
    else if (s == "cd_pass_ratio") h->cd_pass_ratio = (int)value;   // each further pass stops at ratio x the previous limit
    else if (s == "list_fine") h->list_fine = (int)value;       // 1 (default) = per-entry statistics on v_mfma_f64_4x4x4 for 16 <= K <= 31, 0 = on 16x16x4
    else if (s == "q_split") h->q_split = (int)value;             // 0 = both products in one piece behind the row phase
    else if (s == "join_lean") h->join_lean = (int)value;         // experiment of round 5, off: bits 1 | 2 gain 0.6 % at c3 and cost c2 2.5 % (chained joins add their wake-up latencies)
    else if (s == "mm_fast") h->mm_fast = (int)value;             // 0 = k_mm_rows / k_mm_reduce as in round 4
    else if (s == "col_mfma4") h->col_mfma4 = (int)value;         // 1 = k_col_paircnt4 (K <= 31, factor rows fit LDS), 0 = k_col_paircnt
    else if (s == "cd_pairs") h->cd_pairs = (int)value;           // 1 (default) = sweeps routed through the blocks of two coordinate steps (K <= 30; same iterates), 0 = one step per block
    else if (s == "cd_split") h->cd_split = (int)value;           // 2 = steady-state column steps run split (long genes first, on their own stream); 0 (default) = never (measured: no gain, see use_split)
    else if (s == "cd_long_frac") h->cd_long_frac = value;        // at most this fraction of the genes counts as long (default 0.03)
    else if (s == "cd_variant") h->cd_variant = (int)value;   // 0 = register-resident (4 genes per wave; K <= 32, and 32 < K <= 48 with the third slot's columns in LDS), 1 = group kernel, 2 = row16 (LDS, K <= 48)
    else return fail(INSIDER_ERR_ARG, "unknown option " + s);
    return INSIDER_OK;
}

static int optimize_body(insider_hip_handle *h, double *const *A, double *C, int inc_continuous, int K, double lambda1,
                         double lambda2, double alpha, int tuning, double global_tol, double sub_tol, uint32_t max_iter,
                         uint64_t seed, double *out_train_rmse, double *out_test_rmse, double *out_loss, double *traj,
                         int traj_cap, int *out_traj_rows, int *out_iters)
{
    int rc = check_factor_args(h, A, C, inc_continuous, tuning);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(h->device));
    // work a failed earlier call may have left on the side streams must not race with this call's
    HIPCHECK(hipStreamSynchronize(h->side));
    HIPCHECK(hipStreamSynchronize(h->side2));
    HIPCHECK(hipStreamSynchronize(h->side3));
    HIPCHECK(hipStreamSynchronize(h->lng));
    if ((rc = ensure_workspace(h, K))) return rc;
    const auto t_begin = std::chrono::steady_clock::now();
    clear_events(h);
    h->w_ready = false;
    h->side_pending = false;
    h->qfull_pending = false;
    h->qheld_pending = false;
    h->q_kb = 0;
    h->long_pending = false;
    const int masked = tuning == 1;
    if ((rc = upload_factors(h, A, C, K))) return rc;

    // ---- fit of the initial values (src/optimize.cpp:320-323) ----------------------------------------------------
    LossOut lo;
    if ((rc = phase_R(h))) return rc;
    if (masked) if ((rc = launch_col_stats(h, false))) return rc;
    if ((rc = launch_col_solve(h, masked, false, lambda2, alpha, sub_tol, 1, false))) return rc;
    if ((rc = launch_test_sse(h, masked, false))) return rc;
    if ((rc = loss_checkpoint(h, tuning, lambda1, lambda2, alpha, &lo))) return rc;
    double loss = lo.loss, pre_loss, decay = 1.0;
    double train_rmse = lo.train_rmse, test_rmse = lo.test_rmse;
    int trows = 0;
    auto put_traj = [&](double it, double delta, double dec) {
        if (traj && trows < traj_cap) {
            double *t = traj + (size_t)trows * INSIDER_TRAJ_STRIDE;
            t[0] = it; t[1] = lo.train_rmse; t[2] = lo.test_rmse; t[3] = lo.sum_residual / 2; t[4] = lo.row_reg_half;
            t[5] = lo.col_reg_half; t[6] = lo.l1_reg; t[7] = lo.loss; t[8] = delta; t[9] = dec;
            ++trows;
        }
    };
    put_traj(-1, std::numeric_limits<double>::quiet_NaN(), decay);

    uint32_t iter = 0;
    unsigned long long sweeps_total = 0;
    HIPCHECK(hipMemsetAsync(h->sweep_total, 0, 256 * sizeof(unsigned long long), h->stream));
    HIPCHECK(hipMemsetAsync(h->failflag + 2, 0, 2 * sizeof(int), h->stream));
    if (alpha != 0.0 && !h->have_perm && !h->have_early[0]) {   // no history on this handle: order the genes by sum of squares
        hipLaunchKernelGGL(k_yy_key, dim3(cdiv(h->p, 256)), dim3(256), 0, h->stream,
                           (const double *)(masked ? h->yy_train : h->yy_all), (int)h->p, h->sweep_key);
        KCHECK();
        if ((rc = launch_gene_order(h, nullptr, 0, 1, h->stream))) return rc;
        h->have_perm = true;
    }
    while (iter <= max_iter) {                                                                  // :325
        if (h->verbose && iter % 10 == 0) printf("Iteration %u ---------------------------------\n", iter);
        // ---- row step: all covariates, Gauss-Seidel (:332-362) -------------------------------------------------
        if (use_merged(h, masked)) { if ((rc = launch_wsyrk_side(h))) return rc; }                // incl. the row prep
        else if ((rc = launch_row_prep(h, masked))) return rc;                                  // :332
        if (masked && !use_merged(h, masked)) if ((rc = launch_row_stats(h, true))) return rc;
        // V = C A' of the covariates 1 .. c-1: what covariate 0's update reads.  Covariate 0's own columns are first read by
        // covariate 1's update, after they have been recomputed from the updated factors (below): not formed here
#ifdef INSIDER_V_ALL   // (A/B builds: every column, as before round 3)
        if (use_merged(h, masked)) if ((rc = launch_gene_v(h, 0, h->SL))) return rc;
#else
        // (with continuous covariates on the merged form: their columns of V too — s_r carries A_c' z_r)
        if (use_merged(h, masked)) if ((rc = launch_gene_v(h, h->c > 1 ? h->lvl_off[1] : h->SLcat, h->SL))) return rc;
#endif
        if (use_merged(h, masked) && h->row_head) HIPCHECK(hipStreamWaitEvent(h->stream, h->ev_head, 0));   // launch_wsyrk_side
        const bool cont_follow = inc_continuous && h->m > 0;
        const int q_kb = use_merged(h, masked) ? q_split_boundary(h, masked, cont_follow) : 0;
        for (int i = 0; i < h->c; ++i) {
            const bool need_R = !(use_merged(h, masked) || unmasked_fused(h, masked)) || (i + 1 == h->c && !cont_follow);
            if ((rc = row_update(h, i, -1, masked, lambda1, need_R))) return rc;                // :339
            if (use_merged(h, masked) && (i + 1 < h->c || cont_follow))   // (the continuous columns read every categorical column of V)
                if ((rc = launch_gene_v(h, h->lvl_off[i], h->lvl_off[i + 1]))) return rc;
            if (q_kb && i + (cont_follow ? 1 : 2) == h->c)   // every block but the last is final: its part of Qfull / Qheld, beside the last update
                if ((rc = launch_q_early(h, q_kb))) return rc;
        }
        if (cont_follow)
            for (int j = 0; j < h->m; ++j) {
                if ((rc = row_update(h, 0, j, masked, lambda1, !use_merged(h, masked) || j + 1 == h->m))) return rc;   // :340-351
                if (use_merged(h, masked) && j + 1 < h->m)
                    if ((rc = launch_gene_v(h, h->SLcat + j, h->SLcat + j + 1))) return rc;
            }
        h->w_ready = false;
        // ---- column step (:365-378) -------------------------------------------------------------------------------
        if ((rc = phase_R(h, true, true, masked != 0))) return rc;
        const int checkpoint = iter % 10 == 0;
        if (alpha != 0.0 && iter == 0)   // later iterations: built on the side stream while the previous solve ran
            if ((rc = ensure_order_table(h, seed, iter, K, h->max_sweeps, h->order_mode, lambda2 * alpha, nullptr, 0))) return rc;
        if (masked) if ((rc = launch_col_stats(h, true, use_split(h, masked, alpha, (int)iter)))) return rc;
        if (alpha != 0.0) {
            h->order = h->order_buf[iter & 1];
            if (iter < max_iter) {   // the next iteration's table, from here on: beside this iteration's solve
                HIPCHECK(hipEventRecord(h->ev_tab, h->stream));
                HIPCHECK(hipStreamWaitEvent(h->side, h->ev_tab, 0));
                if ((rc = ensure_order_table(h, seed, iter + 1, K, h->max_sweeps, h->order_mode, lambda2 * alpha, h->side, (int)((iter + 1) & 1))))
                    return rc;
            }
        }
        if ((rc = launch_col_solve(h, masked, true, lambda2, alpha, sub_tol * decay, checkpoint, true, (int)iter, true))) return rc;  // :376
        if (alpha != 0.0)
            if ((rc = side_close(h))) return rc;
        if (checkpoint) {                                                                       // :381-408
            if ((rc = launch_test_sse(h, masked, true))) return rc;
            pre_loss = loss;
            if ((rc = loss_checkpoint(h, tuning, lambda1, lambda2, alpha, &lo))) return rc;
            if ((rc = check_fail_flag(h))) return rc;
            loss = lo.loss;
            train_rmse = lo.train_rmse;
            test_rmse = lo.test_rmse;
            const double delta_loss = pre_loss - loss;
            if (delta_loss / 1000 <= 1e-6) decay = 1e-6;                                        // :389-403
            else if (delta_loss / 1000 <= 1e-5) decay = 1e-5;
            else if (delta_loss / 1000 <= 1e-4) decay = 1e-4;
            else if (delta_loss / 1000 <= 1e-3) decay = 1e-3;
            else if (delta_loss / 1000 <= 1e-2) decay = 1e-2;
            else if (delta_loss / 1000 <= 1e-1) decay = 1e-1;
            else decay = 1.0;
            put_traj(iter, delta_loss, decay);
            if (h->verbose) {
                printf("insider iter %u: train rmse = %.12g\n", iter, train_rmse);
                if (tuning == 1) printf("insider iter %u: test rmse = %.12g\n", iter, test_rmse);
                printf("total_residual\t%.12g;\nrow_reg_loss:\t%.12g;\ncol_reg_loss:\t%.12g;\nl1_reg_loss:\t%.12g.\n",
                       lo.sum_residual / 2, lo.row_reg_half, lo.col_reg_half, lo.l1_reg);
                printf("Delta loss for iter %u:%.12g\n", iter, delta_loss);
            }
            if ((pre_loss - loss) / pre_loss < global_tol) break;                               // :405-407
        }
        ++iter;
    }
    if ((rc = download_factors(h, A, C, K))) return rc;
    if ((rc = check_fail_flag(h))) return rc;
    if ((rc = read_cap_hits(h))) return rc;
    HIPCHECK(hipStreamSynchronize(h->side2));
    HIPCHECK(hipStreamSynchronize(h->side3));
    HIPCHECK(hipStreamSynchronize(h->side));   // the gene orders kept for the next call
    h->side_pending = false;
    {
        unsigned long long bins[256];
        // (on the handle's own stream: a null-stream copy would wait for every other handle's work, insider_hip_clone)
        HIPCHECK(hipMemcpyAsync(bins, h->sweep_total, sizeof(bins), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        for (unsigned long long v : bins) sweeps_total += v;
    }
    if (out_train_rmse) *out_train_rmse = train_rmse;
    if (out_test_rmse) *out_test_rmse = test_rmse;
    if (out_loss) *out_loss = loss;
    if (out_traj_rows) *out_traj_rows = trows;
    if (out_iters) *out_iters = (int)iter;
    // ---- profile -------------------------------------------------------------------------------------------------------
    const auto t_end = std::chrono::steady_clock::now();
    for (double &v : h->prof) v = 0;
    auto sum_events = [&](std::vector<hipEvent_t> &ev, double *launches, double *ms) {
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) { *ms += t; *launches += 1; }
        }
    };
    sum_events(h->ev_col, &h->prof[0], &h->prof[1]);
    sum_events(h->ev_row, &h->prof[2], &h->prof[3]);
    sum_events(h->ev_cd, &h->prof[4], &h->prof[5]);
    sum_events(h->ev_test, &h->prof[6], &h->prof[7]);
    auto tail_mean = [&](std::vector<hipEvent_t> &ev) {   // one event pair per outer iteration: the mean from iteration 5 on
        double ms = 0.0;
        int cnt = 0;
        for (size_t i = 10; i + 1 < ev.size(); i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) { ms += t; ++cnt; }
        }
        return cnt ? ms / cnt : 0.0;
    };
    h->steady_cd_ms = tail_mean(h->ev_cd);
    h->steady_col_ms = tail_mean(h->ev_col);
    h->prof[8] = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    h->prof[9] = (double)std::min<uint64_t>((uint64_t)iter + 1, (uint64_t)max_iter + 1);
    h->prof[10] = (double)sweeps_total;
    h->prof[11] = (masked && use_col_factored(h) ? 1.0 : 0.0) + (use_merged(h, masked) ? 2.0 : 0.0) +
                  (masked && col_stats_path(h) == 2 ? 4.0 : 0.0);   // which statistics paths ran
    clear_events(h);
    return INSIDER_OK;
}

int insider_hip_optimize(insider_hip_handle *h, double *const *A, double *C, int inc_continuous, int K, double lambda1,
                         double lambda2, double alpha, int tuning, double global_tol, double sub_tol, uint32_t max_iter,
                         uint64_t seed, double *out_train_rmse, double *out_test_rmse, double *out_loss, double *traj,
                         int traj_cap, int *out_traj_rows, int *out_iters)
{
    const int rc = optimize_body(h, A, C, inc_continuous, K, lambda1, lambda2, alpha, tuning, global_tol, sub_tol, max_iter,
                                 seed, out_train_rmse, out_test_rmse, out_loss, traj, traj_cap, out_traj_rows, out_iters);
    if (rc != INSIDER_OK && h && h->stream) {
        // an early return leaves enqueued work on all three streams: drain them so that the next call starts clean
        const std::string keep = g_err;
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->side);
        (void)hipStreamSynchronize(h->side2);
        (void)hipStreamSynchronize(h->side3);
        (void)hipStreamSynchronize(h->lng);
        h->side_pending = h->qfull_pending = h->qheld_pending = h->w_ready = h->long_pending = false;
        h->q_kb = 0;
        if (h->failflag) (void)hipMemset(h->failflag, 0, 4 * sizeof(int));
        clear_events(h);
        g_err = keep;
    }
    return rc;
}

int insider_hip_optimize_oneshot_ex(const double *X, int64_t n, int64_t p, double *const *A, double *C,
                                    const int32_t *levels, int c, const int32_t *n_levels, const double *ctns, int m,
                                    const uint8_t *M_train, const uint8_t *M_test, int inc_continuous, int K,
                                    double lambda1, double lambda2, double alpha, int tuning, double global_tol,
                                    double sub_tol, uint32_t max_iter, uint64_t seed, int device, double *out_train_rmse,
                                    double *out_test_rmse, double *out_loss)
{
    if (inc_continuous != 0 && inc_continuous != 1)   // src/optimize.cpp:270-272
        return fail(INSIDER_ERR_ARG, "The value of prarameter inc_continuous can only be 0 or 1.");
    if (inc_continuous == 1 && (!ctns || m < 1))
        return fail(INSIDER_ERR_ARG, "inc_continuous = 1 needs ctns_confounder (n x m, m >= 1)");
    insider_hip_handle *h = nullptr;
    // the reference ignores ctns_confounder when inc_continuous = 0 (src/optimize.cpp:276-291): so does the upload
    int rc = insider_hip_create_ex(X, n, p, levels, c, n_levels, inc_continuous ? ctns : nullptr, inc_continuous ? m : 0,
                                   M_train, M_test, device, &h);
    if (rc) return rc;
    rc = insider_hip_optimize(h, A, C, inc_continuous, K, lambda1, lambda2, alpha, tuning, global_tol, sub_tol, max_iter,
                              seed, out_train_rmse, out_test_rmse, out_loss, nullptr, 0, nullptr, nullptr);
    const std::string keep = g_err;
    insider_hip_destroy(h);
    g_err = keep;
    return rc;
}

int insider_hip_optimize_oneshot(const double *X, int64_t n, int64_t p, double *const *A, double *C,
                                 const int32_t *levels, int c, const int32_t *n_levels, const uint8_t *M_train,
                                 const uint8_t *M_test, int inc_continuous, int K, double lambda1, double lambda2,
                                 double alpha, int tuning, double global_tol, double sub_tol, uint32_t max_iter,
                                 uint64_t seed, double *out_train_rmse, double *out_test_rmse, double *out_loss)
{
    if (inc_continuous == 1)
        return fail(INSIDER_ERR_ARG, "continuous covariates need insider_hip_optimize_oneshot_ex (ctns_confounder)");
    return insider_hip_optimize_oneshot_ex(X, n, p, A, C, levels, c, n_levels, nullptr, 0, M_train, M_test, inc_continuous,
                                           K, lambda1, lambda2, alpha, tuning, global_tol, sub_tol, max_iter, seed, 0,
                                           out_train_rmse, out_test_rmse, out_loss);
}

// One row update of one covariate, the reference's optimize_row() as optimize() calls it (src/optimize.cpp:339 with
// :139-198): the residual is X minus the contributions of every OTHER covariate (their A_i as passed), A[cov] is
// replaced by the per-level ridge solutions.  lambda = 0 on an interaction covariate with the other factors zero is
// fit_interaction()'s arithmetic (src/fit_interaction.cpp:10-90).  cov >= c addresses continuous column cov - c
// (optimize_continuous_v2, :76-137).
int insider_hip_optimize_row(insider_hip_handle *h, double *const *A, const double *C, int inc_continuous, int K, int cov,
                             double lambda, int tuning)
{
    int rc = check_factor_args(h, A, C, inc_continuous, tuning);
    if (rc) return rc;
    if (cov < 0 || cov >= h->c + (inc_continuous ? h->m : 0)) return fail(INSIDER_ERR_ARG, "covariate index out of range");
    if (!(lambda == lambda)) return fail(INSIDER_ERR_ARG, "lambda is NaN");   // any finite value, like the reference
    HIPCHECK(hipSetDevice(h->device));
    if ((rc = ensure_workspace(h, K))) return rc;
    h->w_ready = false;
    if ((rc = upload_factors(h, A, C, K))) return rc;
    if ((rc = launch_row_prep(h, tuning))) return rc;
    if (tuning == 1 && !use_merged(h, tuning)) if ((rc = launch_row_stats(h, false))) return rc;
    if ((rc = launch_build_R(h))) return rc;
    if (use_merged(h, tuning)) if ((rc = launch_gene_v(h, 0, h->SL))) return rc;
    if (cov < h->c) rc = row_update(h, cov, -1, tuning, lambda);
    else rc = row_update(h, 0, cov - h->c, tuning, lambda);
    if (rc) return rc;
    if ((rc = download_factors(h, A, nullptr, K))) return rc;
    return check_fail_flag(h);
}

// One column update, the reference's optimize_col() as optimize() calls it (src/optimize.cpp:376 with :200-253):
// every gene's elastic-net (alpha > 0, warm start C) or ridge (alpha == 0) regression of X on the row factor built
// from A.  `iter` selects the sweep-order stream (include/insider_perm.h).
int insider_hip_optimize_col(insider_hip_handle *h, double *const *A, double *C, int inc_continuous, int K, double lambda,
                             double alpha, int tuning, double tol, uint64_t seed, uint32_t iter)
{
    int rc = check_factor_args(h, A, C, inc_continuous, tuning);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(h->device));
    if ((rc = ensure_workspace(h, K))) return rc;
    if ((rc = upload_factors(h, A, C, K))) return rc;
    if ((rc = phase_R(h))) return rc;
    if (alpha != 0.0) {
        if ((rc = ensure_order_table(h, seed, iter, K, h->max_sweeps, h->order_mode, lambda * alpha, nullptr, 0))) return rc;
        h->order = h->order_buf[0];
    }
    if (tuning == 1) if ((rc = launch_col_stats(h, false))) return rc;
    HIPCHECK(hipMemsetAsync(h->failflag + 2, 0, 2 * sizeof(int), h->stream));
    if ((rc = launch_col_solve(h, tuning, true, lambda, alpha, tol, 0, false))) return rc;
    if ((rc = download_factors(h, nullptr, C, K))) return rc;
    if ((rc = read_cap_hits(h))) return rc;
    return check_fail_flag(h);
}

// device part shared by the two strong_coordinate_descent entries: dG / dq / dw hold nprob problems on `device`
static int strong_cd_device(DevBufs &bufs, const double *dG, const double *dq, const double *dw, int K, int64_t nprob,
                            double lambda, double alpha, double tol, uint64_t seed, uint32_t iter, int order_mode,
                            int max_sweeps, double *beta_out, int32_t *sweeps_out)
{
    double *db = nullptr;
    int *ds = nullptr;
    int rc;
    if ((rc = bufs.alloc(&db, (size_t)nprob * K)) || (rc = bufs.alloc(&ds, (size_t)nprob))) return rc;
    const int ms = max_sweeps < 1 ? 1 : max_sweeps;
    const int rows = std::min<int64_t>(ms, INSIDER_PERM_PERIOD);   // one period of the order sequence (include/insider_perm.h)
    uint8_t *dord = nullptr;
    if ((rc = bufs.alloc(&dord, (size_t)(rows + 4) * ORDER_ROW))) return rc;   // + the look-ahead row
    // debugging knob: INSIDER_CD_VARIANT=2 runs the LDS-resident row16 solver instead of the register-resident one, 1 the group kernel
    const char *var = std::getenv("INSIDER_CD_VARIANT");
    const int variant = var ? std::atoi(var) : 0;
    const bool reg3 = reg3_path(K, lambda * alpha, variant);
    unsigned long long code_base = 0;
    if (reg_pairs(reg_kmax(K)) && variant == 0 && lambda * alpha > 0.0) {   // the register-resident batch kernel will run: where are its code blocks?
        unsigned long long *dcb = nullptr;
        if ((rc = bufs.alloc(&dcb, 1))) return rc;
        HIPCHECK(hipMemset(dcb, 0, sizeof(unsigned long long)));
        CdParams none{};
        REG_DISPATCH(K, hipLaunchKernelGGL((k_cd_batch_reg<SL_, KM_>), dim3(1), dim3(64), 0, 0, dG, dq, dw, K, (int64_t)0, none, db, ds, dcb));
        KCHECK();
        HIPCHECK(hipMemcpy(&code_base, dcb, sizeof(code_base), hipMemcpyDeviceToHost));
        if (!code_base) return fail(INSIDER_ERR_HIP, "the sweep kernel did not publish the address of its code blocks");
    }
    hipLaunchKernelGGL(k_order_table, dim3(cdiv((int64_t)(rows + 1) * 64, 256)), dim3(256), 0, 0, seed, iter, K, rows, order_mode, K * 8,
                       reg_kmax(K), (K > 32 && !reg3) ? 1 : 0, code_base, 0ull, dord);
    KCHECK();
    CdParams cd;
    cd.lambda = lambda;
    cd.alpha = alpha;
    cd.tol = tol;
    cd.la = lambda * alpha;
    cd.l2 = lambda * (1.0 - alpha);
    cd.two_la = 2.0 * cd.la;
    cd.inv_two_la = cd.la > 0.0 ? 0.5 / cd.la : 0.0;
    cd.max_sweeps = ms;
    cd.order = dord;
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    bufs.events.push_back(e0);
    HIPCHECK(hipEventCreate(&e1));
    bufs.events.push_back(e1);
    HIPCHECK(hipEventRecord(e0, 0));
    const bool lds_variant = variant == 2;
    const size_t r16_bytes = (size_t)r16_lds_doubles(K) * sizeof(double);
    if (K <= 16 && lds_variant)
        hipLaunchKernelGGL((k_cd_batch_r16<1>), dim3(cdiv(nprob, 4)), dim3(64), r16_bytes, 0, dG, dq, dw, K, nprob, cd, db, ds);
    else if (K <= 32 && lds_variant)
        hipLaunchKernelGGL((k_cd_batch_r16<2>), dim3(cdiv(nprob, 4)), dim3(64), r16_bytes, 0, dG, dq, dw, K, nprob, cd, db, ds);
    else if (K <= 32 && cd.la > 0.0) {   // (the register-resident solver's state is scaled by 1 / (2 lambda alpha))
        REG_DISPATCH(K, hipLaunchKernelGGL((k_cd_batch_reg<SL_, KM_>), dim3(cdiv(nprob, 4)), dim3(64), 0, 0, dG, dq, dw, K,
                                           nprob, cd, db, ds, (unsigned long long *)nullptr));
    } else if (K <= 16) hipLaunchKernelGGL((k_cd_batch<16, 4>), dim3(cdiv(nprob, 16)), dim3(256), 0, 0, dG, dq, dw, K, nprob, cd, db, ds);
    else if (K <= 32) hipLaunchKernelGGL((k_cd_batch<32, 2>), dim3(cdiv(nprob, 4)), dim3(128), 0, 0, dG, dq, dw, K, nprob, cd, db, ds);
    else if (reg3) {   // 32 < K <= 48: register-resident with the third slot's matrix columns in LDS
        REG3_DISPATCH(K, hipLaunchKernelGGL((k_cd_batch_reg<SL_, KM_>), dim3(cdiv(nprob, 4)), dim3(64), 0, 0, dG, dq, dw, K,
                                            nprob, cd, db, ds, (unsigned long long *)nullptr));
    }
    else if (K <= 48 && variant != 1) {   // lambda alpha = 0 or INSIDER_CD_VARIANT=2: the LDS-resident row16 solver (=1: one problem per wavefront)
        if (int rl = r16_wide_lds(r16_bytes)) return rl;
        hipLaunchKernelGGL((k_cd_batch_r16<3>), dim3(cdiv(nprob, 4)), dim3(64), r16_bytes, 0, dG, dq, dw, K, nprob, cd, db, ds);
    }
    else hipLaunchKernelGGL((k_cd_batch<64, 1>), dim3((unsigned)nprob), dim3(64), 0, 0, dG, dq, dw, K, nprob, cd, db, ds);
    KCHECK();
    HIPCHECK(hipEventRecord(e1, 0));
    HIPCHECK(hipDeviceSynchronize());
    float msf = 0;
    (void)hipEventElapsedTime(&msf, e0, e1);
    g_last_cd_ms = msf;
    HIPCHECK(hipMemcpy(beta_out, db, (size_t)nprob * K * sizeof(double), hipMemcpyDeviceToHost));
    if (sweeps_out) HIPCHECK(hipMemcpy(sweeps_out, ds, (size_t)nprob * sizeof(int), hipMemcpyDeviceToHost));
    return INSIDER_OK;
}

static int cd_common_checks(int K, int64_t nprob, int device)
{
    if (K < 1 || K > 64 || nprob < 0) return fail(INSIDER_ERR_ARG, "K must be in 1..64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(INSIDER_ERR_NO_DEVICE, "no HIP device visible: libinsider_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(INSIDER_ERR_ARG, "bad device ordinal");
    return INSIDER_OK;
}

int insider_hip_strong_cd(const double *XtX, const double *Xty, const double *wstart, int K, int64_t nprob, double lambda,
                          double alpha, double tol, uint64_t seed, uint32_t iter, int order_mode,
                          int max_sweeps, int device, double *beta_out, int32_t *sweeps_out)
{
    if (!XtX || !Xty || !wstart || !beta_out) return fail(INSIDER_ERR_ARG, "null argument");
    int rc = cd_common_checks(K, nprob, device);
    if (rc) return rc;
    if (nprob == 0) return INSIDER_OK;
    HIPCHECK(hipSetDevice(device));
    DevBufs bufs;
    double *dG = nullptr, *dq = nullptr, *dw = nullptr;
    if ((rc = bufs.alloc(&dG, (size_t)nprob * K * K)) || (rc = bufs.alloc(&dq, (size_t)nprob * K)) ||
        (rc = bufs.alloc(&dw, (size_t)nprob * K)))
        return rc;
    HIPCHECK(hipMemcpy(dG, XtX, (size_t)nprob * K * K * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dq, Xty, (size_t)nprob * K * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dw, wstart, (size_t)nprob * K * sizeof(double), hipMemcpyHostToDevice));
    return strong_cd_device(bufs, dG, dq, dw, K, nprob, lambda, alpha, tol, seed, iter, order_mode, max_sweeps, beta_out,
                            sweeps_out);
}

int insider_hip_strong_cd_xy(const double *X, const double *y, int64_t m, int K, const double *wstart, double lambda,
                             double alpha, const double *XtX, const double *Xty, double tol, uint64_t seed, uint32_t iter,
                             int order_mode, int max_sweeps, int device, double *beta_out, int32_t *sweeps_out)
{
    if (!wstart || !beta_out) return fail(INSIDER_ERR_ARG, "null argument");
    if ((!XtX || !Xty) && (!X || !y)) return fail(INSIDER_ERR_ARG, "pass (X, y), or XtX and Xty, or all four");
    if (m < 0) return fail(INSIDER_ERR_ARG, "m must be >= 0");
    int rc = cd_common_checks(K, 1, device);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(device));
    DevBufs bufs;
    double *dG = nullptr, *dq = nullptr, *dw = nullptr;
    if ((rc = bufs.alloc(&dG, (size_t)K * K)) || (rc = bufs.alloc(&dq, (size_t)K)) || (rc = bufs.alloc(&dw, (size_t)K)))
        return rc;
    if (!XtX || !Xty) {   // X'X and X'y on the device from the design matrix and outcome (src/optimize.cpp:219-222,234-235)
        double *dX = nullptr, *dy = nullptr;
        if ((rc = bufs.alloc(&dX, (size_t)m * K)) || (rc = bufs.alloc(&dy, (size_t)m))) return rc;
        HIPCHECK(hipMemcpy(dX, X, (size_t)m * K * sizeof(double), hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dy, y, (size_t)m * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_xtx_xty, dim3(K, K + 1), dim3(64), 0, 0, (const double *)dX, (const double *)dy, m, K, dG, dq);
        KCHECK();
    }
    if (XtX) HIPCHECK(hipMemcpy(dG, XtX, (size_t)K * K * sizeof(double), hipMemcpyHostToDevice));
    if (Xty) HIPCHECK(hipMemcpy(dq, Xty, (size_t)K * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dw, wstart, (size_t)K * sizeof(double), hipMemcpyHostToDevice));
    return strong_cd_device(bufs, dG, dq, dw, K, 1, lambda, alpha, tol, seed, iter, order_mode, max_sweeps, beta_out,
                            sweeps_out);
}

int insider_hip_solve_sympd(const double *A, const double *b, int K, int64_t nsys, int device, double *x, int32_t *route)
{
    if (!A || !b || !x) return fail(INSIDER_ERR_ARG, "null argument");
    int rc = cd_common_checks(K, nsys, device);
    if (rc) return rc;
    if (nsys == 0) return INSIDER_OK;
    HIPCHECK(hipSetDevice(device));
    DevBufs bufs;
    double *dA = nullptr, *db = nullptr, *dx = nullptr;
    int *dr = nullptr;
    if ((rc = bufs.alloc(&dA, (size_t)nsys * K * K)) || (rc = bufs.alloc(&db, (size_t)nsys * K)) ||
        (rc = bufs.alloc(&dx, (size_t)nsys * K)) || (rc = bufs.alloc(&dr, (size_t)nsys)))
        return rc;
    HIPCHECK(hipMemcpy(dA, A, (size_t)nsys * K * K * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(db, b, (size_t)nsys * K * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_solve_batch, dim3((unsigned)nsys), dim3(64), 0, 0, (const double *)dA, (const double *)db, K, nsys, dx,
                       dr);
    KCHECK();
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(x, dx, (size_t)nsys * K * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<int> hr(nsys);
    HIPCHECK(hipMemcpy(hr.data(), dr, (size_t)nsys * sizeof(int), hipMemcpyDeviceToHost));
    bool singular = false;
    for (int64_t i = 0; i < nsys; ++i) {
        if (route) route[i] = hr[i];
        singular = singular || hr[i] < 0;
    }
    if (singular) return fail(INSIDER_ERR_SOLVE, "a system is singular to working precision");
    return INSIDER_OK;
}

// optimize_continuous_v2 (src/optimize.cpp:76-137) with the reference's eight arguments, on an arbitrary `data` matrix
// (insider_cont_v2.hpp): one streaming pass over (data, indicator) for the per-gene sums, the K x K weighted Gram, then the
// reference's cyclic scalar passes (tuning = 1, k_cont_cd) or its one ridge solve (tuning = 0, k_level_solve).
int insider_hip_optimize_continuous_v2(const double *data, int64_t n, int64_t p, const uint8_t *indicator,
                                       double *updating_factor, const double *c_factor, int K, const double *updating_confd,
                                       const double *gram, double lambda, int tuning, int device)
{
    if (tuning != 0 && tuning != 1)   // the reference prints and exit(1)s (src/optimize.cpp:133-136)
        return fail(INSIDER_ERR_ARG, "Parameter tuning should be either 0 or 1!");
    if (!data || !updating_factor || !c_factor || !updating_confd) return fail(INSIDER_ERR_ARG, "null argument");
    if (tuning == 1 && !indicator) return fail(INSIDER_ERR_ARG, "tuning = 1 needs the indicator matrix");
    if (tuning == 0 && !gram) return fail(INSIDER_ERR_ARG, "tuning = 0 needs gram (K x K)");
    if (n < 1 || p < 1) return fail(INSIDER_ERR_ARG, "n and p must be >= 1");
    if (!(lambda == lambda)) return fail(INSIDER_ERR_ARG, "lambda is NaN");
    if (K < 1 || K > INSIDER_MAX_K) return fail(INSIDER_ERR_UNSUPPORTED, "K must be in 1..63");
    int rc = cd_common_checks(K, 1, device);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(device));
    const int NB = (K + 1 + 15) / 16, KP = 16 * NB, len = KP * KP + KP;
    const int nslab = cdiv(p, CV2_SLAB);
    const size_t np = (size_t)n * (size_t)p;
    DevBufs bufs;
    double *dD = nullptr, *dC = nullptr, *dz = nullptr, *dw = nullptr, *dt = nullptr, *dpart = nullptr, *deq = nullptr, *du = nullptr,
           *dg = nullptr, *dzz = nullptr;
    uint8_t *dM = nullptr;
    int *dflag = nullptr;   // [0] the solve's fail flag, [1] a level count of 1 for k_level_solve
    if ((rc = bufs.alloc(&dD, np)) || (rc = bufs.alloc(&dC, (size_t)K * p)) || (rc = bufs.alloc(&dz, (size_t)n)) ||
        (rc = bufs.alloc(&dt, (size_t)p)) || (rc = bufs.alloc(&dpart, (size_t)nslab * len)) || (rc = bufs.alloc(&deq, (size_t)len)) ||
        (rc = bufs.alloc(&du, (size_t)KP)) || (rc = bufs.alloc(&dflag, 2)))
        return rc;
    HIPCHECK(hipMemcpy(dD, data, np * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dC, c_factor, (size_t)K * p * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dz, updating_confd, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(du, 0, (size_t)KP * sizeof(double)));
    HIPCHECK(hipMemcpy(du, updating_factor, (size_t)K * sizeof(double), hipMemcpyHostToDevice));
    const int flag0[2] = {0, 1};
    HIPCHECK(hipMemcpy(dflag, flag0, sizeof(flag0), hipMemcpyHostToDevice));
    if (tuning == 1) {
        if ((rc = bufs.alloc(&dM, np)) || (rc = bufs.alloc(&dw, (size_t)p))) return rc;
        HIPCHECK(hipMemcpy(dM, indicator, np, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k_cv2_gene<true>), dim3(cdiv(p, 4)), dim3(256), 0, 0, (const double *)dD, (const uint8_t *)dM,
                           (const double *)dz, n, p, dw, dt);
    } else {
        if ((rc = bufs.alloc(&dg, (size_t)K * K)) || (rc = bufs.alloc(&dzz, 1))) return rc;
        HIPCHECK(hipMemcpy(dg, gram, (size_t)K * K * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_cv2_zz, dim3(1), dim3(64), 0, 0, (const double *)dz, n, dzz);
        hipLaunchKernelGGL((k_cv2_gene<false>), dim3(cdiv(p, 4)), dim3(256), 0, 0, (const double *)dD, (const uint8_t *)nullptr,
                           (const double *)dz, n, p, (double *)nullptr, dt);
    }
    KCHECK();
    hipLaunchKernelGGL(k_cv2_eq_part, dim3(nslab), dim3(256), 0, 0, (const double *)dC, (const double *)dw, (const double *)dt, K, KP,
                       p, dpart);
    KCHECK();
    hipLaunchKernelGGL(k_cv2_eq_sum, dim3(cdiv(len, 256)), dim3(256), 0, 0, (const double *)dpart, nslab, K, KP,
                       (const double *)dg, (const double *)dzz, deq);
    KCHECK();
    NB_DISPATCH(NB, {
        (void)WPB_;
        if (tuning == 1)   // :102-126
            hipLaunchKernelGGL((k_cont_cd<NB_>), dim3(1), dim3(64), 0, 0, (const double *)deq, K, lambda, du);
        else               // :127-131
            hipLaunchKernelGGL((k_level_solve<NB_>), dim3(1), dim3(64), 0, 0, (const double *)deq, (const int *)(dflag + 1), 1, K,
                               lambda, du, dflag);
    });
    KCHECK();
    HIPCHECK(hipDeviceSynchronize());
    int flag = 0;
    HIPCHECK(hipMemcpy(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) return fail(INSIDER_ERR_SOLVE, "the ridge system of the continuous covariate is singular to working precision");
    HIPCHECK(hipMemcpy(updating_factor, du, (size_t)K * sizeof(double), hipMemcpyDeviceToHost));
    return INSIDER_OK;
}

static int masked_gram_common(insider_hip_handle *h, bool cols, const double *Fhost, int K, double *G_out, double *q_out)
{
    if (!h || !Fhost || !G_out || !q_out) return fail(INSIDER_ERR_ARG, "null argument");
    HIPCHECK(hipSetDevice(h->device));
    int rc = ensure_workspace(h, K);
    if (rc) return rc;
    const int KP = h->KP;
    const int64_t units = cols ? h->p : h->n, flen = cols ? h->n : h->p;
    double *F = cols ? h->R : h->C, *full = cols ? h->RtR : h->CCt;
    // host factor: cols -> R is n x K column-major; rows -> C is K x p column-major (= p rows of K)
    HIPCHECK(hipMemcpy(h->stage, Fhost, (size_t)flen * K * sizeof(double), hipMemcpyHostToDevice));
    if (cols) hipLaunchKernelGGL(k_pack_A, dim3(cdiv(flen * KP, 256)), dim3(256), 0, h->stream, (const double *)h->stage,
                                 (int)flen, K, KP, F);
    else hipLaunchKernelGGL(k_pack_rows, dim3(cdiv(flen * KP, 256)), dim3(256), 0, h->stream, (const double *)h->stage,
                            flen, K, KP, F);
    KCHECK();
    if ((rc = launch_gram(h, F, flen, full))) return rc;
    double *stat = nullptr, *qf = nullptr, *Gd = nullptr, *qd = nullptr;
    const int NBLK = h->NB * (h->NB + 1) / 2, STAT = NBLK * 256;
    const int nseg = cols ? 1 : h->nseg;
    DevBufs bufs;
    if ((rc = bufs.alloc(&stat, (size_t)nseg * units * STAT)) || (rc = bufs.alloc(&qf, (size_t)units * KP)) ||
        (rc = bufs.alloc(&Gd, (size_t)units * K * K)) || (rc = bufs.alloc(&qd, (size_t)units * K)))
        return rc;
    if ((rc = launch_list_stats(h, cols, nseg, F, stat))) return rc;
    // dense X'F over all entries from the gene-major copy (rows: strided reads; stand-alone API only)
    if (cols) hipLaunchKernelGGL(k_line_dense_xty, dim3((unsigned)units), dim3(64), 0, h->stream, (const double *)h->X,
                                 h->ldn, (int)flen, (const double *)F, K, KP, qf);
    else hipLaunchKernelGGL(k_row_dense_xty, dim3((unsigned)units), dim3(64), 0, h->stream, (const double *)h->X, h->ldn,
                            (int)flen, (const double *)F, K, KP, qf);
    KCHECK();
    NB_DISPATCH(h->NB, {
        (void)WPB_;
        hipLaunchKernelGGL((k_stats_to_dense<NB_>), dim3((unsigned)units), dim3(64), 0, h->stream, (const double *)stat,
                           nseg, (int)units, K, (const double *)full, (const double *)qf, Gd, qd);
    });
    KCHECK();
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipMemcpy(G_out, Gd, (size_t)units * K * K * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(q_out, qd, (size_t)units * K * sizeof(double), hipMemcpyDeviceToHost));
    // the pad rows of C (genes p..ldp-1) were not touched; R/C now hold the caller's factor
    return INSIDER_OK;
}

int insider_hip_masked_gram_cols(insider_hip_handle *h, const double *R, int K, double *G_out, double *q_out)
{
    return masked_gram_common(h, true, R, K, G_out, q_out);
}

int insider_hip_masked_gram_rows(insider_hip_handle *h, const double *C, int K, double *H_out, double *b_out)
{
    return masked_gram_common(h, false, C, K, H_out, b_out);
}

double insider_hip_last_cd_ms(void) { return g_last_cd_ms; }

int insider_hip_get_sweeps(insider_hip_handle *h, int32_t *out)
{
    if (!h || !out) return fail(INSIDER_ERR_ARG, "null");
    if (!h->sweeps) return fail(INSIDER_ERR_ARG, "no column update has run yet");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipMemcpy(out, h->sweeps, (size_t)h->p * sizeof(int), hipMemcpyDeviceToHost));
    return INSIDER_OK;
}

int insider_hip_get_info(insider_hip_handle *h, const char *name, double *out)
{
    if (!h || !name || !out) return fail(INSIDER_ERR_ARG, "null");
    const std::string s(name);
    const int NB = h->NB;
    if (s == "col_stats_path") *out = col_stats_path(h);
    else if (s == "row_merged") *out = use_merged(h, 1) ? 1.0 : 0.0;
    else if (s == "col_entries") *out = (double)h->col_entries;      // padded held-out list entries, column side
    else if (s == "row_entries") *out = (double)h->row_entries;
    else if (s == "stat_doubles") *out = NB ? NB * (NB + 1) / 2 * 256.0 : 0.0;
    else if (s == "kp") *out = h->KP;
    else if (s == "pair_count_bytes_per_gene") *out = h->cf_pair_ok ? h->cf.cnt_stride : 0.0;
    else if (s == "lists_bytes") *out = 12.0 * ((double)h->col_entries + (double)h->row_entries);
    else if (s == "cap_hits") *out = h->cap_hits;                   // last optimize() / optimize_col(): solves ended by max_sweeps
    else if (s == "max_gene_sweeps") *out = h->max_gene_sweeps;     // ... and the longest solve, in sweeps
    else if (s == "max_sweeps") *out = h->max_sweeps;
    else if (s == "cd_ms_steady") *out = h->steady_cd_ms;           // option "profile": mean over outer iterations >= 5 of the last call
    else if (s == "col_stats_ms_steady") *out = h->steady_col_ms;
    else if (s == "col_mfma_per_gene") {
        // v_mfma_f64_16x16x4_f64 instructions the column-side statistics kernel issues per gene (2048 flops each; the 4x4x4 form
        // of the per-entry kernel is counted in the same unit: a quarter per instruction)
        if (!NB) return fail(INSIDER_ERR_ARG, "no workspace yet: run an update first");
        const int path = col_stats_path(h);
        double v = 0.0;
        if (path == 0) {
            const int NT = (h->K + 4) / 4;
            if (h->list_fine && NB == 2 && NT >= 5 && NT <= 8 && h->fperm)   // k_list_stats4: NT (NT + 1) / 2 instructions of 512 flops per 16 entries
                v = (double)h->col_entries / (double)std::max<int64_t>(h->p, 1) / 16.0 * (NT * (NT + 1) / 2) / 4.0;
            else
                v = (double)h->col_entries / (double)std::max<int64_t>(h->p, 1) / 4.0 * (NB * (NB + 1) / 2);
        } else
            for (int t = 0; t < h->cf.c; ++t) {
                v += std::ceil(h->cf.L[t] / 4.0) * NB * NB;                                           // M += A' P
                if (path == 2 && h->cf.nlater[t] > 0) v += std::ceil(h->cf.L[t] / 16.0) * h->cf.nsteps * NB;   // P = N_j Tab
                if (path == 2 && h->m > 0) v += std::ceil(h->cf.L[t] / 16.0) * NB;                            // + real-valued counts
            }
        if (path == 2 && h->m > 0) v += std::ceil(h->m / 4.0) * NB * NB + NB;                                 // the continuous position
        *out = v;
    } else return fail(INSIDER_ERR_ARG, "unknown info key " + s);
    return INSIDER_OK;
}

int insider_hip_get_array(insider_hip_handle *h, const char *name, void *out, int64_t bytes)
{
    if (!h || !name || !out) return fail(INSIDER_ERR_ARG, "null");
    const std::string s(name);
    const void *src = nullptr;
    int64_t have = 0;
    if (s == "cd_pass_slot") { src = h->cd_pass_slot; have = h->p * (int64_t)sizeof(int); }
    else if (s == "gene_perm") { src = h->gene_perm; have = h->p * (int64_t)sizeof(int); }
    else if (s == "order_table") { src = h->order; have = (int64_t)(h->order_rows + 1) * ORDER_ROW; }   // rows of ORDER_ROW bytes: the last solve's
    else return fail(INSIDER_ERR_ARG, "unknown array " + s);
    if (!src || bytes > have) return fail(INSIDER_ERR_ARG, "array not available or too short");
    HIPCHECK(hipSetDevice(h->device));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipMemcpy(out, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return INSIDER_OK;
}

int insider_hip_get_profile(insider_hip_handle *h, double *out12)
{
    if (!h || !out12) return fail(INSIDER_ERR_ARG, "null");
    for (int i = 0; i < 12; ++i) out12[i] = h->prof[i];
    return INSIDER_OK;
}

}  // extern "C"
