// ubench2.hip — do integer VALU instructions of one wave overlap with f64 MFMAs of another wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ inline d4 dz() { d4 z; z[0] = 0; z[1] = 0; z[2] = 0; z[3] = 0; return z; }
#define REPS 4000
// mode bit0: waves with (wave index / 4) even run MFMA loops; bit1: odd ones run integer VALU loops (else idle)
__global__ void k_mix(int mode, double seed, unsigned long long *out)
{
    const int wave = threadIdx.x >> 6;     // 8 waves per block -> 2 per SIMD
    const bool second = wave >= 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!second) {
        if (mode & 1) {
            d4 c0 = dz(), c1 = dz(), c2 = dz(), c3 = dz();
            double a = seed + threadIdx.x, b = 1.0 + threadIdx.x;
            for (int r = 0; r < REPS; ++r) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
            if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) out[1000] = 1;
        }
    } else {
        if (mode & 2) {
            int a0 = 1, a1 = 2, a2 = 3, a3 = 4, b = (int)seed;
            for (int r = 0; r < REPS * 4; ++r)
                asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n"
                             "v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
            if (a0 + a1 + a2 + a3 == 12345678) out[1000] = 1;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 1 << 20);
    for (int mode : {1, 2, 3, 1, 2, 3}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, mode, 1.5, d);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (1=mfma only: %d mfma/wave, 2=valu only: %d valu/wave, 3=both): %.3f ms  ticks mfma-wave %llu valu-wave %llu\n",
               mode, REPS * 4, REPS * 32, ms, h[0], h[4]);
    }
    return 0;
}
