// Issue rate of v_mfma_f64_4x4x4 (four 4x4x4 blocks per instruction, 512 flop) against v_mfma_f64_16x16x4 (2048 flop) on MI355X,
// one wave per SIMD, eight independent accumulator chains; and the lane layout of the 4x4x4 form's operands and result.
// hipcc --offload-arch=gfx950 -O2 -o ubench_mfma4 tools/ubench_mfma4.hip && ./ubench_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_small(double *out, int n)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void __launch_bounds__(256) k_big(double *out, int n)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
// The same two loops with OPERANDS THAT DIFFER from instruction to instruction and carry full random mantissas (round 5: the
// constant-operand loops above run the 4x4x4 form at 0.93 of nominal, the real statistics kernel built on it runs at 0.62 like the
// 16x16x4 one: is the difference the data?)
__device__ __forceinline__ double rnd(unsigned &st)
{
    st = st * 1664525u + 1013904223u;
    const unsigned hi = st;
    st = st * 1664525u + 1013904223u;
    return __hiloint2double((int)((hi & 0x000fffffu) | 0x3fe00000u), (int)st);   // [0.5, 1), 52 random mantissa bits
}
__global__ void __launch_bounds__(256) k_small_var(double *out, int n)
{
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x;
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(st) - 0.75; b[i] = rnd(st) - 0.75; }
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], b[3], c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[1], c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], b[2], c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], b[3], c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], b[0], c7, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void __launch_bounds__(256) k_big_var(double *out, int n)
{
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x;
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd(st) - 0.75; b[i] = rnd(st) - 0.75; }
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
// layout: A = indicator of one (lane) element, B = indicator of one lane element -> which output lanes light up
__global__ void k_layout(double *out)
{
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        // A nonzero only in lane la (value 1), B = 1 + lane index: the result shows which B lanes pair with A's lane and where it lands
        const double a = l == la ? 1.0 : 0.0, b = 1.0 + l;
        out[la * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    }
}
int main()
{
    double *out;
    hipMalloc(&out, 4 * 256 * 256 * sizeof(double) + 64 * 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    for (int WPS = 1; WPS <= 4; ++WPS) {   // waves per SIMD (blocks of four waves per CU)
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); k_small<<<dim3(256 * WPS), 256>>>(out, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("4x4x4 (4 blocks), %d wave(s) per SIMD: %.3f ms: %.2f ns per instruction and SIMD = %.1f flop/ns/SIMD\n", WPS, ms, ms * 1e6 / (n * 8.0 * WPS), 512.0 / (ms * 1e6 / (n * 8.0 * WPS)));
        hipEventRecord(e0); k_big<<<dim3(256 * WPS), 256>>>(out, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4         , %d wave(s) per SIMD: %.3f ms: %.2f ns per instruction and SIMD = %.1f flop/ns/SIMD\n", WPS, ms, ms * 1e6 / (n * 4.0 * WPS), 2048.0 / (ms * 1e6 / (n * 4.0 * WPS)));
    }
    }
    for (int WPS = 1; WPS <= 4; ++WPS) {
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); k_small_var<<<dim3(256 * WPS), 256>>>(out, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("4x4x4 random operands, %d wave(s) per SIMD: %.3f ms: %.2f ns per instruction and SIMD = %.1f flop/ns/SIMD\n", WPS, ms, ms * 1e6 / (n * 8.0 * WPS), 512.0 / (ms * 1e6 / (n * 8.0 * WPS)));
        hipEventRecord(e0); k_big_var<<<dim3(256 * WPS), 256>>>(out, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4 random operands, %d wave(s) per SIMD: %.3f ms: %.2f ns per instruction and SIMD = %.1f flop/ns/SIMD\n", WPS, ms, ms * 1e6 / (n * 4.0 * WPS), 2048.0 / (ms * 1e6 / (n * 4.0 * WPS)));
    }
    }
    static double h[64 * 64];
    k_layout<<<1, 64>>>(out);
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; la += 1) {
        if (!(la < 8 || la % 16 == 0 || la == 21)) continue;
        printf("A lane %2d ->", la);
        for (int l = 0; l < 64; ++l) if (h[la * 64 + l] != 0.0) printf(" out[%d]=B[%d]", l, (int)h[la * 64 + l] - 1);
        printf("\n");
    }
    return 0;
}
