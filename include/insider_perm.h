/*
 * insider_perm.h — deterministic, counter-based sweep order for the elastic-net
 * coordinate descent.
 *
 * The reference draws a fresh uniformly random order for every sweep with
 * Armadillo's randperm() (src/coordinate_descent.cpp:89), which under
 * RcppArmadillo pulls from R's global, non-thread-safe RNG from inside an
 * OpenMP region (src/optimize.cpp:213-228): the order is unreproducible even
 * on the reference itself.  Both the CPU oracle and the HIP kernels therefore
 * take the order from this header instead: coordinate l of the active set gets
 * the 32-bit key below and the sweep visits active coordinates in ascending
 * key order.  The low 6 bits of a key are the coordinate index, so keys are
 * unique (K <= 64) and the order is a pure function of
 * (seed, outer iteration, sweep counter).  It does NOT depend on the gene: the
 * subproblems are independent, so sharing one fresh random order per sweep
 * number across genes leaves every gene's own order sequence uniformly random
 * (the reference's distribution) while letting the HIP kernel run several
 * genes per wavefront in lock-step and read the order from a small
 * precomputed table.  Sweeping "all coordinates in key order, skipping the
 * inactive ones" is the same as a uniformly random order over the active set
 * (src/coordinate_descent.cpp:89-92).
 *
 * Pure uint32 wrap-around arithmetic: bit-identical in gcc and in hipcc
 * device code.
 */
#ifndef INSIDER_PERM_H
#define INSIDER_PERM_H

#include <stdint.h>

#if defined(__HIPCC__)
#define INSIDER_HD __host__ __device__ __forceinline__
#else
#define INSIDER_HD static inline
#endif

/* "lowbias32" integer finaliser (public-domain constants). */
INSIDER_HD uint32_t insider_h32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

/* The order sequence of one solve repeats after INSIDER_PERM_PERIOD sweeps (a power of two): sweep s takes the order of
 * sweep s mod PERIOD.  A solve that long has seen 16384 fresh uniformly random orders; what the period buys is a
 * sweep-order table of fixed size on the device (PERIOD rows per outer iteration), so that the number of sweeps of a solve
 * needs no cap — the reference's loop has none (src/coordinate_descent.cpp:86-114). */
#ifndef INSIDER_PERM_PERIOD   /* (tests/test_gpu_period.py builds library AND oracle with 64u to cross the wrap cheaply) */
#define INSIDER_PERM_PERIOD 16384u
#endif

/* Per-(seed, iteration, sweep) base word; uniform across coordinates and genes. */
INSIDER_HD uint32_t insider_perm_base(uint64_t seed, uint32_t iter, uint32_t sweep)
{
    sweep &= INSIDER_PERM_PERIOD - 1U;
    uint32_t b = insider_h32((uint32_t)seed ^ 0x9E3779B9U);
    b = insider_h32(b ^ (uint32_t)(seed >> 32) ^ (0x85EBCA6BU * iter));
    b = insider_h32(b + 0xC2B2AE35U * sweep);
    return b;
}

/* Sort key of coordinate l (0 <= l < 64). */
INSIDER_HD uint32_t insider_perm_key(uint32_t base, uint32_t l)
{
    return (insider_h32(base ^ (0x27D4EB2FU * (l + 1U))) & 0xFFFFFFC0U) | l;
}

#endif /* INSIDER_PERM_H */
