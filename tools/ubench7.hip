// ubench7.hip — does the computed jump's cost depend on how many instruction-cache lines a block touches?  A shortened
// block (6 VALU + movrels + M0 add + setpc = 60 bytes) chained through the SGPR table at 64-byte spacing aligned to
// the 64-byte line (one line per block), at 64-byte spacing offset by 32 bytes (two lines per block), and falling through.
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
    "s_add_u32 m0, m0, 1\n"                                      \
    "v_fmac_f64_dpp %[h0], %[dn], %[g0] row_newbcast:" #IT " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp %[h1], %[dn], %[g1] row_newbcast:" #IT " row_mask:0xf bank_mask:0xf\n"
#define JB(N, IT, OFF) ".org Lc%= + " STR(OFF) " + 64*" #N "\n" BLK(IT) "s_setpc_b64 vcc\n"
#define ADDR(K, OFF) "s_add_u32 s" #K ", s98, " STR(OFF) " + 64*(" #K "-64)\n"
#define OPS : [h0] "+v"(h0), [h1] "+v"(h1), [b0] "+v"(b0), [c] "=&v"(c), [dn] "=&v"(dn) \
            : [i0] "v"(i0), [g0] "v"(g0), [g1] "v"(g1), [la] "s"(la) \
            : "vcc", "scc", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s98", "s99"
#define CHAIN(OFF)                                                                                                \
    asm volatile("s_mov_b32 s63, m0\n s_getpc_b64 s[98:99]\n Lr%=:\n s_add_u32 s98, s98, Lc%=-Lr%=\n s_addc_u32 s99, s99, 0\n" \
                 ADDR(64, OFF) ADDR(65, OFF) ADDR(66, OFF) ADDR(67, OFF) ADDR(68, OFF) ADDR(69, OFF) ADDR(70, OFF) ADDR(71, OFF) \
                 ADDR(72, OFF) ADDR(73, OFF) ADDR(74, OFF) ADDR(75, OFF) ADDR(76, OFF) ADDR(77, OFF) ADDR(78, OFF) ADDR(79, OFF) ADDR(80, OFF) \
                 "s_mov_b32 m0, 1\n s_mov_b32 vcc_hi, s99\n s_mov_b32 vcc_lo, s64\n s_setpc_b64 vcc\n .p2align 12\n Lc%=:\n"  \
                 JB(0, 0, OFF) JB(1, 1, OFF) JB(2, 2, OFF) JB(3, 3, OFF) JB(4, 4, OFF) JB(5, 5, OFF) JB(6, 6, OFF) JB(7, 7, OFF) \
                 JB(8, 8, OFF) JB(9, 9, OFF) JB(10, 10, OFF) JB(11, 11, OFF) JB(12, 12, OFF) JB(13, 13, OFF) JB(14, 14, OFF) JB(15, 15, OFF) \
                 ".org Lc%= + " STR(OFF) " + 64*16\n s_mov_b32 m0, s63\n" OPS)

template <int T>
__global__ void __launch_bounds__(1024) k(double seed, double la, double *out)
{
    double h0 = seed + threadIdx.x, h1 = h0 + 1, b0 = 0.25, i0 = 0.5, g0 = 1e-3, g1 = 2e-3, c, dn;
    for (int r = 0; r < REPS; ++r) {
        if (T == 0)
            asm volatile("s_mov_b32 s63, m0\n s_mov_b32 m0, 0\n" BLK(0) BLK(1) BLK(2) BLK(3) BLK(4) BLK(5) BLK(6) BLK(7) BLK(8) BLK(9) BLK(10)
                         BLK(11) BLK(12) BLK(13) BLK(14) BLK(15) "s_mov_b32 m0, s63\n" OPS);
        if (T == 1) CHAIN(0);
        if (T == 2) CHAIN(32);
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
        printf("%-44s waves/SIMD %d: %.3f ms  %.2f ns per block per SIMD\n", name, wps, ms, ms * 1e6 / ((double)REPS * 16 * wps));
    }
}

int main()
{
    double *d;
    (void)hipMalloc(&d, 1 << 16);
    run<0>("60-byte block, fall through", d);
    run<1>("60-byte block + setpc, one line per block", d);
    run<2>("60-byte block + setpc, two lines per block", d);
    return 0;
}
