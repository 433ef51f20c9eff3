// ubench4.hip — which part of the register-resident CD block (insider_cd_reg.hpp) bounds its saturated rate?
// Variants of the block, 16 copies in a loop: full / no exec writes / no movrels+M0 / chained with s_branch / s_setpc.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 500
#define VCHAIN                                                   \
    "v_min_f64 %[c], %[h0], %[la]\n"                             \
    "v_max_f64 %[c], %[c], -%[la]\n"                             \
    "v_add_f64 %[c], %[h0], -%[c]\n"                             \
    "v_fma_f64 %[dn], -%[c], %[i0], %[b0]\n"                     \
    "v_mul_f64 %[c], %[c], %[i0]\n"
#define FM "v_fmac_f64_dpp %[h0], %[dn], %[g0] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
           "v_fmac_f64_dpp %[h1], %[dn], %[g1] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define EXECW "s_lshl_b64 exec, %[lm], 3\n v_mov_b64 %[b0], %[c]\n s_mov_b64 exec, %[ex]\n"
#define NOEXECW "s_nop 0\n v_mov_b64 %[b0], %[c]\n s_nop 0\n"
#define MREL "s_movrels_b32 vcc_lo, s64\n"
#define M0INC "s_add_u32 m0, m0, 0\n"
#define B_FULL MREL VCHAIN EXECW M0INC FM
#define B_NOEXEC MREL VCHAIN NOEXECW M0INC FM
#define B_NOMREL VCHAIN EXECW FM
#define B_VALU VCHAIN "v_mov_b64 %[b0], %[c]\n" FM
#define B_BRANCH(N) MREL VCHAIN EXECW M0INC FM "s_branch Lq" #N "_%=\n s_nop 0\n s_nop 0\n s_nop 0\n Lq" #N "_%=:\n"
#define B_SETPC(N)  MREL VCHAIN EXECW M0INC FM "s_add_u32 s98, s98, 96\n s_addc_u32 s99, s99, 0\n s_setpc_b64 s[98:99]\n .org Lc%= + 96*" #N "\n"
#define X16(B) B B B B B B B B B B B B B B B B
#define OPS : [h0] "+v"(h0), [h1] "+v"(h1), [b0] "+v"(b0), [c] "=&v"(c), [dn] "=&v"(dn), [ex] "=&s"(ex) \
            : [i0] "v"(i0), [g0] "v"(g0), [g1] "v"(g1), [la] "s"(la), [lm] "s"(lm) : "vcc", "scc", "s64", "s65", "s98", "s99"

template <int T>
__global__ void __launch_bounds__(1024) k(double seed, double la, double *out)
{
    double h0 = seed + threadIdx.x, h1 = h0 + 1, b0 = 0.25, i0 = 0.5, g0 = 1e-3, g1 = 2e-3, c, dn;
    unsigned long long ex;
    const unsigned long long lm = 0x0001000100010001ull;
    for (int r = 0; r < REPS; ++r) {
        if (T == 0) asm volatile("s_mov_b64 %[ex], exec\n s_mov_b32 m0, 0\n" X16(B_FULL) OPS);
        if (T == 1) asm volatile("s_mov_b64 %[ex], exec\n s_mov_b32 m0, 0\n" X16(B_NOEXEC) OPS);
        if (T == 2) asm volatile("s_mov_b64 %[ex], exec\n" X16(B_NOMREL) OPS);
        if (T == 3) asm volatile("s_mov_b64 %[ex], exec\n" X16(B_VALU) OPS);
        if (T == 4)
            asm volatile("s_mov_b64 %[ex], exec\n s_mov_b32 m0, 0\n" B_BRANCH(0) B_BRANCH(1) B_BRANCH(2) B_BRANCH(3) B_BRANCH(4) B_BRANCH(5)
                         B_BRANCH(6) B_BRANCH(7) B_BRANCH(8) B_BRANCH(9) B_BRANCH(10) B_BRANCH(11) B_BRANCH(12) B_BRANCH(13) B_BRANCH(14)
                         B_BRANCH(15) OPS);
        if (T == 5)
            asm volatile("s_mov_b64 %[ex], exec\n s_mov_b32 m0, 0\n s_getpc_b64 s[98:99]\n Lr%=:\n s_add_u32 s98, s98, Lc%=-Lr%=\n s_addc_u32 s99, s99, 0\n"
                         "s_setpc_b64 s[98:99]\n .p2align 12\n Lc%=:\n .org Lc%= + 96*0\n"
                         B_SETPC(1) B_SETPC(2) B_SETPC(3) B_SETPC(4) B_SETPC(5) B_SETPC(6) B_SETPC(7) B_SETPC(8) B_SETPC(9) B_SETPC(10)
                         B_SETPC(11) B_SETPC(12) B_SETPC(13) B_SETPC(14) B_SETPC(15) B_SETPC(16) OPS);
    }
    if (h0 + h1 + b0 == 12345.678) out[0] = 1;
}

template <int T>
void run(const char *name, double *d)
{
    for (int wps : {1, 2, 3, 4}) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(256 * wps), 0, 0, 1.5, 0.75, d);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(256 * wps), 0, 0, 1.5, 0.75, d);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s waves/SIMD %d: %.3f ms  %.2f ns per block per SIMD  (%.2f ns per block per wave)\n", name, wps, ms,
               ms * 1e6 / ((double)REPS * 16 * wps), ms * 1e6 / ((double)REPS * 16));
    }
}

int main()
{
    double *d;
    (void)hipMalloc(&d, 1 << 16);
    run<3>("VALU only (8 VALU)", d);
    run<2>("VALU + exec writes", d);
    run<1>("VALU + movrels + M0 add (no exec writes)", d);
    run<0>("full block, fall through", d);
    run<4>("full block + taken s_branch", d);
    run<5>("full block + s_add/s_addc/s_setpc to next", d);
    return 0;
}
