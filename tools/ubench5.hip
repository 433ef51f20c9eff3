// ubench5.hip — cost of the computed jump of the register-resident CD block (insider_cd_reg.hpp) versus block spacing:
// the kernel's own 7-VALU block, 16 blocks chained through an SGPR address table (s_movrels + s_setpc), blocks
// STRIDE bytes apart; and the same block falling through.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 500
#define XSTR(x) #x
#define STR(x) XSTR(x)
#define BLK(IT)                                                 \
    "s_movrels_b32 vcc_lo, s64\n"                                \
    "v_min_f64 %[c], %[h0], %[la]\n"                             \
    "v_max_f64 %[c], %[c], -%[la]\n"                             \
    "v_add_f64 %[c], %[h0], -%[c]\n"                             \
    "v_fma_f64 %[dn], -%[c], %[i0], %[b0]\n"                     \
    "s_lshl_b64 exec, %[lm], " #IT "\n"                          \
    "s_add_u32 m0, m0, 1\n"                                      \
    "v_add_f64 %[b0], %[b0], -%[dn]\n"                           \
    "s_mov_b64 exec, %[ex]\n"                                    \
    "v_fmac_f64_dpp %[h0], %[dn], %[g0] row_newbcast:" #IT " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %[h1], %[dn], %[g1] row_newbcast:" #IT " row_mask:0xf bank_mask:0xf\n"
#define JB(N, IT, STRIDE) ".org Lc%= + " STR(STRIDE) "*" #N "\n" BLK(IT) "s_setpc_b64 vcc\n"
#define FT(N, IT) BLK(IT)
#define ADDR(K, STRIDE) "s_add_u32 s" #K ", s98, " STR(STRIDE) "*(" #K "-64)\n"
#define OPS : [h0] "+v"(h0), [h1] "+v"(h1), [b0] "+v"(b0), [c] "=&v"(c), [dn] "=&v"(dn), [ex] "=&s"(ex) \
            : [i0] "v"(i0), [g0] "v"(g0), [g1] "v"(g1), [la] "s"(la), [lm] "s"(lm) \
            : "vcc", "scc", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s98", "s99"
#define CHAIN(STRIDE, ALIGN)                                                                                                \
    asm volatile("s_mov_b64 %[ex], exec\n s_mov_b32 s63, m0\n s_getpc_b64 s[98:99]\n Lr%=:\n s_add_u32 s98, s98, Lc%=-Lr%=\n s_addc_u32 s99, s99, 0\n" \
                 ADDR(64, STRIDE) ADDR(65, STRIDE) ADDR(66, STRIDE) ADDR(67, STRIDE) ADDR(68, STRIDE) ADDR(69, STRIDE) ADDR(70, STRIDE) ADDR(71, STRIDE) \
                 ADDR(72, STRIDE) ADDR(73, STRIDE) ADDR(74, STRIDE) ADDR(75, STRIDE) ADDR(76, STRIDE) ADDR(77, STRIDE) ADDR(78, STRIDE) ADDR(79, STRIDE) ADDR(80, STRIDE) \
                 "s_mov_b32 m0, 1\n s_mov_b32 vcc_hi, s99\n s_mov_b32 vcc_lo, s64\n s_setpc_b64 vcc\n .p2align " #ALIGN "\n Lc%=:\n"  \
                 JB(0, 0, STRIDE) JB(1, 1, STRIDE) JB(2, 2, STRIDE) JB(3, 3, STRIDE) JB(4, 4, STRIDE) JB(5, 5, STRIDE) JB(6, 6, STRIDE) JB(7, 7, STRIDE) \
                 JB(8, 8, STRIDE) JB(9, 9, STRIDE) JB(10, 10, STRIDE) JB(11, 11, STRIDE) JB(12, 12, STRIDE) JB(13, 13, STRIDE) JB(14, 14, STRIDE) JB(15, 15, STRIDE) \
                 ".org Lc%= + " STR(STRIDE) "*16\n s_mov_b32 m0, s63\n" OPS)

template <int T>
__global__ void __launch_bounds__(1024) k(double seed, double la, double *out)
{
    double h0 = seed + threadIdx.x, h1 = h0 + 1, b0 = 0.25, i0 = 0.5, g0 = 1e-3, g1 = 2e-3, c, dn;
    unsigned long long ex;
    const unsigned long long lm = 0x0001000100010001ull;
    for (int r = 0; r < REPS; ++r) {
        if (T == 0)
            asm volatile("s_mov_b64 %[ex], exec\n s_mov_b32 s63, m0\n s_mov_b32 m0, 0\n" FT(0, 0) FT(1, 1) FT(2, 2) FT(3, 3) FT(4, 4) FT(5, 5) FT(6, 6) FT(7, 7)
                         FT(8, 8) FT(9, 9) FT(10, 10) FT(11, 11) FT(12, 12) FT(13, 13) FT(14, 14) FT(15, 15) "s_mov_b32 m0, s63\n" OPS);
        if (T == 1) CHAIN(96, 12);
        if (T == 2) CHAIN(128, 12);
        if (T == 3) CHAIN(192, 12);
        if (T == 4) CHAIN(256, 13);
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
    run<0>("7-VALU block, fall through", d);
    run<1>("movrels + setpc, 96 B blocks", d);
    run<2>("movrels + setpc, 128 B blocks", d);
    run<3>("movrels + setpc, 192 B blocks", d);
    run<4>("movrels + setpc, 256 B blocks", d);
    return 0;
}
