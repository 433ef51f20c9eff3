/*
 * insider_hip.h — C ABI of libinsider_hip.so, the MI355X-native INSIDER
 * factorisation core.
 *
 * This is the drop-in boundary for the reference's Rcpp exports
 * (/root/reference/src/RcppExports.cpp:112-124, R/RcppExports.R:4-22): plain
 * pointers and sizes, no SEXP / Rcpp / Armadillo / torch types.  All host
 * matrices are column-major fp64 exactly as R hands them to the reference
 * (zero-copy views, src/optimize.cpp:283-284); masks are uint8 instead of the
 * reference's fp64 copies (src/RcppExports.cpp:96-97); level ids are int32,
 * 1-based, exactly 1..L_i per covariate (src/optimize.cpp:175,286 index rows as
 * level-1; validated here, status INSIDER_ERR_ARG otherwise).
 *
 * Error model: every entry point returns an int status (0 = ok) and never
 * calls exit() (the reference does on bad `tuning`, src/optimize.cpp:249-251,
 * 270-272); insider_hip_last_error() returns the message of the calling
 * thread's last failure.  The library fails loudly (INSIDER_ERR_NO_DEVICE)
 * when no HIP device is present: there is no CPU fallback.
 */
#ifndef INSIDER_HIP_H
#define INSIDER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INSIDER_OK 0
#define INSIDER_ERR_ARG 1        /* bad argument (tuning not in {0,1}, level ids not 1..L_i, K out of range, ...) */
#define INSIDER_ERR_SOLVE 2      /* a normal-equation system is singular to working precision (neither route of
                                    solve(..., likely_sympd) could solve it) */
#define INSIDER_ERR_ALLOC 3      /* host or device allocation failed */
#define INSIDER_ERR_HIP 4        /* HIP runtime error (message has the call) */
#define INSIDER_ERR_NO_DEVICE 5  /* no HIP device / extension unusable: no fallback exists */
#define INSIDER_ERR_UNSUPPORTED 6 /* K > 63, n or p >= 2^23, ... */
#define INSIDER_ERR_COMM 7       /* RCCL or the all-reduce callback reported failure, or world > 1 has neither */

/* Largest latent dimension the kernels support (K + 1 augmented column <= 64). */
#define INSIDER_MAX_K 63

/* Number of doubles per trajectory row written by insider_hip_optimize():
 * {iter, train_rmse, test_rmse, SSE/2, row_reg/2, col_reg/2, l1_reg, loss, delta_loss, decay}.
 * Row 0 is the evaluation of the initial values (iter = -1; src/optimize.cpp:320-323); one row per
 * checkpoint follows (iter % 10 == 0; src/optimize.cpp:381-408).  These are the quantities the reference
 * prints to stdout (src/utils.cpp:70-76,95-100). */
#define INSIDER_TRAJ_STRIDE 10

typedef struct insider_hip_handle insider_hip_handle;

/* Sum-all-reduce of `count` doubles at device pointer `dev_buf`, in place, across the gene-sharded ranks.
 * Called by insider_hip_optimize() on the calling thread.  `stream` is the library's HIP stream (a hipStream_t):
 * the kernels producing dev_buf have been ENQUEUED on it, not necessarily completed, and the consumers will be
 * enqueued on it after the call returns.  The callback must therefore order the reduction after the prior work of
 * `stream` and before its later work — either by enqueueing the collective on / against that stream (no host
 * synchronisation needed; insider_amd/dist.py does this through torch.cuda.ExternalStream), or by synchronising
 * the stream, reducing, and synchronising again.  Return 0 on success. */
typedef int (*insider_allreduce_fn)(void *user, double *dev_buf, int64_t count, void *stream);

/* "insider_hip <version> (gfx950) src:<sha16>": the hash is over the sources the library was compiled from
 * (insider_amd/csrc, include/, compiler flags; insider_amd/_build.py), so a caller can tell which sources a number belongs to. */
const char *insider_hip_version(void);
const char *insider_hip_last_error(void);
/* Number of visible HIP devices (0 if none); does not create a context. */
int insider_hip_device_count(void);

/*
 * Upload one data set (or one gene slab of it) to HBM and precompute everything that does not depend on the
 * factors: the combined uint8 mask codes, transposed copies for the row-side pass, per-level row sums of X,
 * per-gene sums of squares.  Replaces the per-call marshaling of src/RcppExports.cpp:91-97 — tune()'s grid
 * (R/insider.R:142-174) re-uploads nothing but the inits.
 *   X        n x p column-major fp64 (NA entries must hold 0, R/insider.R:26)
 *   levels   n x c column-major int32, 1-based ids (cfd_indicators, src/optimize.cpp:256)
 *   n_levels c entries, L_i
 *   M_train  n x p uint8 (train_indicator), M_test n x p uint8 (test_indicator); an entry with both 0 is NA
 *   device   HIP device ordinal
 */
int insider_hip_create(const double *X, int64_t n, int64_t p, const int32_t *levels, int c, const int32_t *n_levels,
                       const uint8_t *M_train, const uint8_t *M_test, int device, insider_hip_handle **out);
/* The same with continuous covariates (ctns_confounder of R/insider.R:48-51; src/optimize.cpp:276-291): ctns is
 * n x m column-major fp64 (NULL / 0 for none).  insider_hip_optimize() must then be called with inc_continuous = 1
 * and c + 1 row-factor pointers, the last one m x K column-major (cfd_matrices(cfd_num-1), :281-291). */
int insider_hip_create_ex(const double *X, int64_t n, int64_t p, const int32_t *levels, int c, const int32_t *n_levels,
                          const double *ctns, int m, const uint8_t *M_train, const uint8_t *M_test, int device,
                          insider_hip_handle **out);
void insider_hip_destroy(insider_hip_handle *h);

/* A second handle on the SAME resident data set: the read-only device arrays insider_hip_create built (X, mask codes,
 * held-out lists, level sums, pair counts: all of it) are shared, the factor workspace, the streams and the options (copied
 * from `src` as they stand) are the clone's own.  Handles of one data set may run insider_hip_optimize() at the same time
 * from different host threads: tune()'s grid points (R/insider.R:142-174) are independent fits of one data set, and a
 * data set of the size real INSIDER inputs have (377 x 5000 ... 44477, README.md:30) does not fill the GPU with one fit.
 * Every handle is destroyed with insider_hip_destroy(); the data set is freed with the last of them.  Results of a fit do
 * not depend on which handle ran it or on what ran beside it. */
int insider_hip_clone(insider_hip_handle *src, insider_hip_handle **out);

/* Gene-axis sharding: this handle holds genes [gene_offset, gene_offset + p) of the global
 * matrix.  gene_offset keys the per-gene sweep order so results do not depend on the sharding.  `fn` (may be
 * NULL when world == 1, or when insider_hip_comm_init() supplies the exchange) is called once per covariate per outer
 * iteration (level normal equations) and once per checkpoint (loss terms).  A non-NULL `fn` replaces a communicator
 * installed earlier by insider_hip_comm_init(). */
int insider_hip_set_shard(insider_hip_handle *h, int64_t gene_offset, int rank, int world, insider_allreduce_fn fn,
                          void *user);

/* In-library RCCL (the collective BASELINE.json's north star names): the per-covariate level equations and the loss
 * terms are summed over the gene-sharded ranks by ncclAllReduce ENQUEUED ON THE LIBRARY'S OWN STREAM, between the kernels
 * that produce and consume them; the callback of insider_hip_set_shard() is then not used (it remains as the fallback
 * for hosts that bring their own communicator).  Rank 0 obtains an id with insider_hip_comm_unique_id(), the host
 * distributes those INSIDER_COMM_ID_BYTES bytes to every rank by any means (insider_amd/dist.py: one broadcast over
 * torch.distributed), and every rank calls insider_hip_comm_init() after insider_hip_set_shard(h, offset, rank, world,
 * NULL, NULL).  Collective call: returns when all `world` ranks have joined.  The communicator is destroyed with the
 * handle. */
#define INSIDER_COMM_ID_BYTES 128
int insider_hip_comm_unique_id(void *out, int out_bytes);
int insider_hip_comm_init(insider_hip_handle *h, const void *unique_id, int rank, int world);

/* Options: "max_sweeps" (safety cap on the sweeps of one elastic-net subproblem, default 2^24: the reference's loop has none,
 * src/coordinate_descent.cpp:86-114, and neither does this library in practice — the sweep-order table holds one period of
 * the order sequence, INSIDER_PERM_PERIOD = 16384 rows, whatever the cap; insider_hip_get_info("cap_hits") counts the solves
 * of the last call that the cap ended), "order_mode" (0 = hashed random order of
 * include/insider_perm.h, 1 = cyclic), "profile" (1 = time the statistics / solve kernels with HIP events),
 * "verbose" (1 = print the reference's per-checkpoint lines to stdout), "cd_variant" (elastic-net sweep kernel:
 * 0 = four genes per wavefront with the Gram matrix in registers [K <= 32; 32 < K <= 48 with the third coordinate slot's
 * columns in LDS], 2 = four genes per wavefront with the Gram matrix in LDS [K <= 48], 1 = one lane group per gene [also
 * what K > 48 takes]; all three follow the same sweep orders and agree to rounding), "row_merged" (1, default = masked
 * row update from per-(level, gene) weighted terms, 0 = from per-sample statistics; same results), "col_factored" (1,
 * default = a cost model picks the form of the column-side masked Gram statistics, 0 = one rank-one update per held-out
 * entry, 2 = per-(covariate, level) terms with one table look-up per entry, 3 = per-(covariate, level) terms from the
 * gene's dense level-pair counts [falls back to 2 when a count exceeds one byte]; same results), "row_counts" (1, default = the merged row update takes its per-gene level sums from the dense
 * level-pair counts when they exist, 0 = from the entry lists; same results), "force_allreduce" (1 = call the all-reduce callback even
 * when world == 1: plumbing rehearsal), "cd_split" / "cd_long_frac" (2 = steady-state column steps run split: the genes predicted longest — whole buckets of the
 * launch order, at most cd_long_frac [0.03] of the genes — get their statistics and their solve on a stream of their own, ahead
 * of the others' statistics; bit-identical results; 0 [default] = off: measured, it does not shorten the step),
 * "row_fused" (1, default = the merged row update forms a level's equations and
 * solve in one launch; with tuning = 0 the whole unmasked row update of a covariate is that one launch; 0 = separate launches),
 * "list_fine" (1, default = the per-entry statistics kernel uses the 4x4x4 form of the f64 matrix instruction for 16 <= K <= 31
 * [fewer wasted outputs than 16 x 16 blocks], 0 = the 16x16x4 form; same sums in another order), "row_gemm" / "row_gemm_waves" (1, default = the per-level weighted Gram sums of a covariate with >= 49
 * levels come from one GEMM over genes, cut into row_gemm_waves [1024] waves; 0 = one weighted rank-one update per (level, gene);
 * same sums in another order, results agree to rounding), "col_mfma4" (1, default = the pair-count column statistics [K <= 31, the
 * factor rows of all covariates within 64 KB of LDS] form sum_l a_l p_l' on the 4x4x4 form of the f64 matrix instruction, rows read
 * in four rotations from LDS, resident blocks whose waves draw genes by ticket; 0 = the 16x16x4 form; same sums to rounding),
 * "mm_fast" (1, default = the streaming products of the row phase [V = C A', S A, U'C, S'C: from 16384 rows on] stage their small
 * operand in LDS once per block and read the tall one in 16-byte pieces / several column tiles per wave; 0 = round 4's kernels;
 * same sums, the row products in another order), "join_lean" and "q_split" (0: experiments of round 5 that gained nothing — fewer
 * stream joins on the main chain [bits 1, 2, 4]; S A and S^held A in two parts, the first beside the last covariate's update),
 * "cd_pairs" (1, default = the register-resident sweep kernel [K <= 30] is routed through its blocks of TWO
 * coordinate steps wherever two consecutive coordinates of a sweep's order share a coordinate slot: a third fewer computed jumps,
 * the same steps in the same order — bit-identical iterates; 0 = one step per block), "cd_pass1" / "cd_pass_ratio" / "cd_cold_iters" (multi-pass column solves in the first
 * cd_cold_iters outer iterations of a call [default 3]: the register-resident sweep kernel stops at sweep cd_pass1 [64; 0 = one
 * pass], cd_pass1 x ratio [4], ..., re-packing the genes still running by their estimated remaining length between passes;
 * the iterates are bit-identical to the single-pass solve). */
int insider_hip_set_option(insider_hip_handle *h, const char *name, double value);

/*
 * The reference's optimize() (src/optimize.cpp:255-422; .Call symbol _insider_optimize,
 * src/RcppExports.cpp:87-110) for categorical covariates.  Same argument meaning:
 *   A            c pointers, A[i] is L_i x K column-major, IN/OUT (the reference mutates cfd_factors in place,
 *                src/optimize.cpp:283-284)
 *   C            K x p column-major, IN/OUT (column_factor, mat&)
 *   lambda1/lambda2/alpha/tuning/global_tol/sub_tol/max_iter as src/optimize.cpp:256-257
 *                (max_iter + 1 outer iterations are run: `iter <= max_iter`, :325)
 *   seed         replaces Rcpp::RNGScope / R's global RNG (src/RcppExports.cpp:90)
 *   out_*        train_rmse, test_rmse (NaN when tuning = 0: uninitialised in the reference, :264), loss
 *   traj         optional, traj_cap rows of INSIDER_TRAJ_STRIDE doubles; out_traj_rows rows written
 *   out_iters    value of `iter` when the loop ended
 * inc_continuous must be 1 exactly when the handle was created with continuous covariates
 * (insider_hip_create_ex); each column j is then updated after the categorical covariates by
 * optimize_continuous_v2 (src/optimize.cpp:76-137,340-351): cyclic scalar CD to sum|du| < 0.1 (tuning = 1) or one
 * ridge solve (tuning = 0).  The one-shot form takes categorical covariates only.
 */
int insider_hip_optimize(insider_hip_handle *h, double *const *A, double *C, int inc_continuous, int K,
                         double lambda1, double lambda2, double alpha, int tuning, double global_tol, double sub_tol,
                         uint32_t max_iter, uint64_t seed, double *out_train_rmse, double *out_test_rmse,
                         double *out_loss, double *traj, int traj_cap, int *out_traj_rows, int *out_iters);

/* One-shot form with the reference's 16 logical arguments (create + optimize + destroy): the direct replacement of
 * .Call(`_insider_optimize`, ...) (src/RcppExports.cpp:87-110, R/RcppExports.R:20-22) that r/insider_hip_shim.c binds.
 *   ctns / m     ctns_confounder (n x m column-major); read only when inc_continuous = 1, exactly as the reference
 *                ignores it otherwise (src/optimize.cpp:276-291); A then carries c + 1 pointers (the last one m x K)
 *   device       HIP device ordinal
 * insider_hip_optimize_oneshot() is the same for categorical covariates on device 0. */
int insider_hip_optimize_oneshot_ex(const double *X, int64_t n, int64_t p, double *const *A, double *C,
                                    const int32_t *levels, int c, const int32_t *n_levels, const double *ctns, int m,
                                    const uint8_t *M_train, const uint8_t *M_test, int inc_continuous, int K,
                                    double lambda1, double lambda2, double alpha, int tuning, double global_tol,
                                    double sub_tol, uint32_t max_iter, uint64_t seed, int device, double *out_train_rmse,
                                    double *out_test_rmse, double *out_loss);
int insider_hip_optimize_oneshot(const double *X, int64_t n, int64_t p, double *const *A, double *C,
                                 const int32_t *levels, int c, const int32_t *n_levels, const uint8_t *M_train,
                                 const uint8_t *M_test, int inc_continuous, int K, double lambda1, double lambda2,
                                 double alpha, int tuning, double global_tol, double sub_tol, uint32_t max_iter,
                                 uint64_t seed, double *out_train_rmse, double *out_test_rmse, double *out_loss);

/*
 * The two block updates as stand-alone operators (the reference's internal optimize_row / optimize_col, reachable
 * there only through optimize()).  Both take the factors in the host layout of insider_hip_optimize.
 *
 * insider_hip_optimize_row — one row update of covariate `cov` (src/optimize.cpp:139-198 as called at :339): the
 *   residual is X minus the contributions of every other covariate as passed in A; A[cov] (L_cov x K) is replaced by
 *   the per-level solutions of (sum_{r in level}(CC' - C_z C_z') + lambda I) a = sum_r C_nz resid[r, nz] (tuning = 1)
 *   or (|level| CC' + lambda I) a = sum_r C resid[r, :]' (tuning = 0).  lambda = 0 with the other
 *   factors zero is fit_interaction()'s arithmetic (src/fit_interaction.cpp:10-90, which applies no ridge term).  The
 *   solve is solve(..., likely_sympd): a system that is not positive definite (lambda <= 0) takes the general route.
 *   cov in [c, c+m) with inc_continuous = 1 updates row cov-c of the continuous factor (optimize_continuous_v2,
 *   src/optimize.cpp:76-137).  Returns INSIDER_ERR_SOLVE when a level's system is singular to working precision.
 * insider_hip_optimize_col — one column update (src/optimize.cpp:200-253 as called at :376): every gene's
 *   elastic-net regression of X[:, j] on the row factor R = sum_i Z_i A_i over its training entries, warm-started
 *   at C[:, j] (alpha > 0), or the ridge solve (alpha == 0); C is updated in place.  `iter` picks the sweep-order
 *   stream of include/insider_perm.h (optimize() passes its outer iteration number).
 */
int insider_hip_optimize_row(insider_hip_handle *h, double *const *A, const double *C, int inc_continuous, int K, int cov,
                             double lambda, int tuning);
int insider_hip_optimize_col(insider_hip_handle *h, double *const *A, double *C, int inc_continuous, int K,
                             double lambda, double alpha, int tuning, double tol, uint64_t seed, uint32_t iter);

/*
 * strong_coordinate_descent (src/coordinate_descent.cpp:56-127; .Call symbol
 * _insider_strong_coordinate_descent, src/RcppExports.cpp:35-50), batched: nprob independent K-variable
 * elastic-net subproblems, one wavefront each, solved in covariance form from (XtX, Xty) — the design
 * matrix X and outcome y of the reference signature enter only through XtX = X'X and Xty = X'y, which the
 * reference's callers always pass alongside (src/optimize.cpp:228,246), so they are not taken here.
 *   XtX    nprob blocks of K x K (column-major; symmetric), Xty / wstart / beta_out nprob blocks of K
 *   seed, iter  key the per-sweep coordinate order (include/insider_perm.h; the same for every subproblem)
 *   sweeps_out  optional, nprob ints
 */
int insider_hip_strong_cd(const double *XtX, const double *Xty, const double *wstart, int K, int64_t nprob,
                          double lambda, double alpha, double tol, uint64_t seed, uint32_t iter, int order_mode,
                          int max_sweeps, int device, double *beta_out, int32_t *sweeps_out);
/* The same solver with the reference's eight arguments, one subproblem (.Call `_insider_strong_coordinate_descent`,
 * src/RcppExports.cpp:35-50: X, y, wstart, lambda, alpha, XtX, Xty, tol): X is m x K column-major, y has m entries.
 * XtX / Xty may be NULL: they are then formed on the device as X'X and X'y (what the reference's callers pass,
 * src/optimize.cpp:219-222,234-235).  When they are given, X and y may be NULL. */
int insider_hip_strong_cd_xy(const double *X, const double *y, int64_t m, int K, const double *wstart, double lambda,
                             double alpha, const double *XtX, const double *Xty, double tol, uint64_t seed, uint32_t iter,
                             int order_mode, int max_sweeps, int device, double *beta_out, int32_t *sweeps_out);

/* solve(A, b, solve_opts::likely_sympd) as the row / ridge updates use it (src/optimize.cpp:175,190,226,240;
 * src/fit_interaction.cpp:54), batched: nsys systems of K x K (column-major) with one right-hand side each.  The
 * positive-definite route (Cholesky) first; when a pivot is not positive, the general route (Gaussian elimination with
 * partial pivoting) — Armadillo's documented behaviour.  route (optional, nsys ints): 0 = Cholesky, 1 = general, -1 =
 * singular (status INSIDER_ERR_SOLVE).  Stand-alone for parity tests of the fallback; the updates call the same
 * device code. */
int insider_hip_solve_sympd(const double *A, const double *b, int K, int64_t nsys, int device, double *x, int32_t *route);

/* optimize_continuous_v2 with the reference's eight arguments (src/optimize.cpp:76-137; .Call symbol
 * `_insider_optimize_continuous_v2`, src/RcppExports.cpp:69-85, R/RcppExports.R:16-18): the update of ONE continuous
 * covariate's K-vector against an arbitrary n x p matrix `data` (inside optimize() it is the Gauss-Seidel residual with this
 * column's own contribution added back, src/optimize.cpp:344-345).
 *   data             n x p column-major
 *   indicator        n x p uint8, non-zero = the entry takes part (train_indicator); read only when tuning = 1
 *   updating_factor  K doubles, IN/OUT (the reference's rowvec&)
 *   c_factor         K x p column-major (column_factor)
 *   updating_confd   n doubles (one column of ctns_confounder)
 *   gram             K x K column-major (column_factor column_factor'); read only when tuning = 0, exactly like the reference
 * tuning = 1: cyclic scalar passes u_i = Xty_i / (XtX_i + lambda) over the masked entries until sum |du| < 0.1 (:102-126);
 * tuning = 0: one solve of ((z'z) gram + lambda I) u = C data' z (:127-131, solve(..., likely_sympd)).  Any other value:
 * INSIDER_ERR_ARG (the reference prints and exit(1)s, :133-136).  Inside a fit the same update runs on the resident data set
 * (insider_hip_optimize / insider_hip_optimize_row with cov >= c); this entry uploads its arguments, runs and frees. */
int insider_hip_optimize_continuous_v2(const double *data, int64_t n, int64_t p, const uint8_t *indicator,
                                       double *updating_factor, const double *c_factor, int K, const double *updating_confd,
                                       const double *gram, double lambda, int tuning, int device);

/*
 * The masked Gram / XtY reductions on their own (for parity tests and profiling).
 * Column side (src/optimize.cpp:216-222): for every gene j, XtX_j = R'R - sum_{i: M_train[i,j]=0} r_i r_i',
 * Xty_j = sum_i M_train[i,j] x_ij r_i.   R is n x K column-major (row_factor).
 *   G_out  p blocks of K x K column-major, q_out p blocks of K.
 * Row side (src/optimize.cpp:162-171 with the data matrix in place of the Gauss-Seidel residual): for every
 * sample r, XtX_r = CC' - sum_{j: M_train[r,j]=0} c_j c_j', Xty_r = sum_j M_train[r,j] x_rj c_j.
 *   C is K x p column-major;  H_out n blocks of K x K, b_out n blocks of K.
 */
int insider_hip_masked_gram_cols(insider_hip_handle *h, const double *R, int K, double *G_out, double *q_out);
int insider_hip_masked_gram_rows(insider_hip_handle *h, const double *C, int K, double *H_out, double *b_out);

/* Profile of the last insider_hip_optimize() call (option "profile" = 1), HIP-event timed on the library's stream.
 * out[0..11]: {column-side masked-Gram launches, total ms, row-side masked-Gram launches, total ms,
 *  column-solve (CD / ridge) launches, total ms, test-residual launches, total ms,
 *  optimize() wall ms, outer iterations run, elastic-net sweeps total, path flags (1 = factored column statistics, 2 =
 *  merged row update, 4 = the factored column statistics ran in their pair-count form)}. */
int insider_hip_get_profile(insider_hip_handle *h, double *out12);

/* Facts about the handle that measurement code needs (bench.py's roofline): "col_stats_path" (0 = per-entry lists, 1 =
 * look-up form, 2 = pair-count form, as the cost model / options chose for the current K), "col_mfma_per_gene"
 * (v_mfma_f64_16x16x4 instructions the column-side statistics kernel issues per gene), "row_merged", "col_entries",
 * "row_entries" (padded held-out list lengths), "lists_bytes", "pair_count_bytes_per_gene", "stat_doubles", "kp",
 * "cd_ms_steady" / "col_stats_ms_steady" (option "profile": mean HIP-event time per outer iteration from iteration 5 on of
 * the last optimize(), i.e. without the cold start), "cap_hits" / "max_gene_sweeps" (of the last optimize() / optimize_col():
 * elastic-net solves ended by "max_sweeps" instead of convergence — must be 0 to match the reference, which has no cap — and
 * the longest solve in sweeps), "max_sweeps". */
int insider_hip_get_info(insider_hip_handle *h, const char *name, double *out);

/* Diagnostics: copy an internal per-gene array to the host: "cd_pass_slot" (uint32 x p: what the last limited pass of a
 * multi-pass column solve left per gene: 0xFFFFFFFF = finished, else estimate bucket << 24 | rank), "gene_perm" (int32 x p:
 * the launch order), "order_table" (the sweep-order table of the last column solve: rows of 448 bytes, one per sweep of the
 * period; bytes 0..K-1 of row s = the coordinates of sweep s in visiting order, include/insider_perm.h). */
int insider_hip_get_array(insider_hip_handle *h, const char *name, void *out, int64_t bytes);

/* Diagnostics: per-gene sweep counts of the last column update (p ints), and the HIP-event time in ms of the
 * kernel launched by the calling THREAD's last insider_hip_strong_cd() / _xy(). */
int insider_hip_get_sweeps(insider_hip_handle *h, int32_t *out);
double insider_hip_last_cd_ms(void);

#ifdef __cplusplus
}
#endif
#endif /* INSIDER_HIP_H */
