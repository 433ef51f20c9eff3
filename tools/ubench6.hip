// ubench6.hip — is a stream of 8-byte fp64 VALU instructions instruction-fetch-bound on gfx950?  Straight-line runs of
// v_fma_f64 (VOP3, 8 B) against v_fmac_f64_e32 (VOP2, 4 B) and v_fmac_f64_dpp (VOP2 + DPP dword, 8 B), four independent
// accumulators, 1..4 waves per SIMD, all CUs or a quarter of them.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 200
#define X4(A) A A A A
#define X16(A) X4(A) X4(A) X4(A) X4(A)
#define X64(A) X16(A) X16(A) X16(A) X16(A)
#define FMA3 "v_fma_f64 %[a0], %[x], %[y], %[a0]\n v_fma_f64 %[a1], %[x], %[y], %[a1]\n v_fma_f64 %[a2], %[x], %[y], %[a2]\n v_fma_f64 %[a3], %[x], %[y], %[a3]\n"
#define FMAC2 "v_fmac_f64_e32 %[a0], %[x], %[y]\n v_fmac_f64_e32 %[a1], %[x], %[y]\n v_fmac_f64_e32 %[a2], %[x], %[y]\n v_fmac_f64_e32 %[a3], %[x], %[y]\n"
#define FMACD "v_fmac_f64_dpp %[a0], %[x], %[y] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[a1], %[x], %[y] row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
              "v_fmac_f64_dpp %[a2], %[x], %[y] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %[a3], %[x], %[y] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define MIN3 "v_min_f64 %[a0], %[a0], %[y]\n v_min_f64 %[a1], %[a1], %[y]\n v_min_f64 %[a2], %[a2], %[y]\n v_min_f64 %[a3], %[a3], %[y]\n"
#define ADD32 "v_add_f32_e32 %[b0], %[b1], %[b0]\n v_add_f32_e32 %[b2], %[b3], %[b2]\n v_add_f32_e32 %[b0], %[b1], %[b0]\n v_add_f32_e32 %[b2], %[b3], %[b2]\n"
#define OPS : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [b0] "+v"(b0), [b2] "+v"(b2) : [x] "v"(x), [y] "v"(y), [b1] "v"(b1), [b3] "v"(b3)

template <int T>
__global__ void __launch_bounds__(1024) k(double seed, double *out)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + threadIdx.x, x = 1e-9, y = 0.5;
    float b0 = 1.f, b1 = 1e-8f, b2 = 2.f, b3 = 1e-9f;
    for (int r = 0; r < REPS; ++r) {
        if (T == 0) asm volatile(X64(FMA3) OPS);
        if (T == 1) asm volatile(X64(FMAC2) OPS);
        if (T == 2) asm volatile(X64(FMACD) OPS);
        if (T == 3) asm volatile(X64(MIN3) OPS);
        if (T == 4) asm volatile(X64(ADD32) OPS);
    }
    if (a0 + a1 + a2 + a3 + b0 + b2 == 12345.678) out[0] = 1;
}

template <int T>
void run(const char *name, double *d)
{
    for (int blocks : {256, 64})
        for (int wps : {1, 2, 4}) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256 * wps), 0, 0, 1.5, d);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256 * wps), 0, 0, 1.5, d);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-36s blocks %3d waves/SIMD %d: %.3f ms  %.3f ns per instruction per SIMD\n", name, blocks, wps, ms,
                   ms * 1e6 / ((double)REPS * 256 * wps));
        }
}

int main()
{
    double *d;
    (void)hipMalloc(&d, 1 << 16);
    run<0>("v_fma_f64 (VOP3, 8 B)", d);
    run<1>("v_fmac_f64_e32 (VOP2, 4 B)", d);
    run<2>("v_fmac_f64_dpp (8 B)", d);
    run<3>("v_min_f64 (VOP3, 8 B)", d);
    run<4>("v_add_f32_e32 (VOP2, 4 B)", d);
    return 0;
}
