// ubench3.hip — issue cost of the instructions of the register-resident CD step on gfx950 (see insider_cd_reg.hpp):
// fp64 VOP3 ops, 64-bit DPP fmac (row_newbcast), v_mov_b64, v_readlane, s_setpc_b64 to an aligned block.
//   hipcc --offload-arch=gfx950 -O3 -o ubench3 tools/ubench3.hip && ./ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 2000
#define R8(X) X X X X X X X X

template <int T>
__global__ void __launch_bounds__(1024) k(double seed, double *out)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0 + 1e-9 * threadIdx.x, c = 0.5;
    int iv = threadIdx.x;
    for (int r = 0; r < REPS; ++r) {
        if (T == 0)
            asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                         "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (T == 1)
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        if (T == 2)
            asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n"
                         "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (T == 3)
            asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n"
                         "v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (T == 4) {
            int s0, s1, s2, s3;
            asm volatile("v_readlane_b32 %0, %4, 1\n v_readlane_b32 %1, %4, 2\n v_readlane_b32 %2, %4, 3\n v_readlane_b32 %3, %4, 4\n"
                         "v_readlane_b32 %0, %4, 5\n v_readlane_b32 %1, %4, 6\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 8\n"
                         : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "v"(iv));
            iv += s0 & 1;
        }
        if (T == 5)   // the dependent chain of one CD step (no dispatch)
            asm volatile("v_min_f64 %1, %0, %3\n v_max_f64 %1, %1, -%3\n v_add_f64 %1, %0, -%1\n v_mul_f64 %1, %1, %4\n v_add_f64 %2, %5, -%1\n"
                         "s_nop 1\n v_fmac_f64_dpp %0, %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(b), "v"(c), "v"(a3));
        if (T == 6)   // chain of 8 computed jumps to 128-byte aligned blocks
            asm volatile("s_getpc_b64 s[96:97]\n Lq%=:\n s_add_u32 s96, s96, Lb%=-Lq%=\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n Lb%=:\n s_add_u32 s96, s96, 128\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n s_add_u32 s96, s96, 128\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n s_add_u32 s96, s96, 128\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n s_add_u32 s96, s96, 128\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n s_add_u32 s96, s96, 128\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n s_add_u32 s96, s96, 128\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n s_add_u32 s96, s96, 128\n s_addc_u32 s97, s97, 0\n s_setpc_b64 s[96:97]\n"
                         ".p2align 7\n" ::: "s96", "s97", "scc");
        if (T == 7)   // v_readlane -> SALU use -> chain (latency of the dispatch arithmetic)
        {
            int s0;
            asm volatile("v_readlane_b32 %0, %1, 1\n s_lshl_b32 %0, %0, 7\n s_add_u32 vcc_lo, %0, 3\n s_addc_u32 vcc_hi, %0, 0\n"
                         "v_readlane_b32 %0, %1, 2\n s_lshl_b32 %0, %0, 7\n s_add_u32 vcc_lo, %0, 3\n s_addc_u32 vcc_hi, %0, 0\n"
                         "v_readlane_b32 %0, %1, 3\n s_lshl_b32 %0, %0, 7\n s_add_u32 vcc_lo, %0, 3\n s_addc_u32 vcc_hi, %0, 0\n"
                         "v_readlane_b32 %0, %1, 4\n s_lshl_b32 %0, %0, 7\n s_add_u32 vcc_lo, %0, 3\n s_addc_u32 vcc_hi, %0, 0\n"
                         : "=&s"(s0) : "v"(iv) : "vcc", "scc");
            iv += s0 & 1;
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + iv == 12345.678) out[0] = 1;
}

template <int T>
void run(const char *name, int per_rep, double *d)
{
    for (int wps : {1, 2, 3, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(256 * wps), 0, 0, 1.5, d);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<T>, dim3(256), dim3(256 * wps), 0, 0, 1.5, d);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD %d: %.3f ms  %.2f ns per item per SIMD  (%.2f ns per item per wave)\n", name, wps, ms,
               ms * 1e6 / ((double)REPS * per_rep * wps), ms * 1e6 / ((double)REPS * per_rep));
    }
}

int main()
{
    double *d;
    hipMalloc(&d, 1 << 16);
    run<0>("v_fma_f64 (independent)", 8, d);
    run<1>("v_fmac_f64_dpp row_newbcast", 8, d);
    run<2>("v_mov_b64", 8, d);
    run<3>("v_min_f64", 8, d);
    run<4>("v_readlane_b32", 8, d);
    run<5>("CD step chain (6 dependent VALU)", 1, d);
    run<6>("s_setpc_b64 to aligned block", 8, d);
    run<7>("readlane + 3 dependent SALU", 4, d);
    return 0;
}
