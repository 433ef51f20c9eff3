// ubench8.hip — how much code can a computed-jump table hold before the jumps start to miss the instruction cache on gfx950?
// N identical 128-byte blocks (6 independent v_fma_f64 + the scalar work that picks the next block) visited in a pseudo-random
// cyclic order (a full-period LCG on the block index, every wave from its own start), 3 waves per SIMD on every CU: time per jump
// against the table's footprint N x 128 B.  Behind DESIGN.md 9 (blocks of two coordinate steps: 62 KB of blocks did not pay).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench8.hip -o tools/ubench8 && tools/ubench8
#include <hip/hip_runtime.h>
#include <cstdio>
#define STR_(x) #x
#define STR(x) STR_(x)
#define JUMPS 20000

#define BODY(N)                                                                                             \
    "s_getpc_b64 s[20:21]\n"                                                                                \
    "Lh%=:\n"                                                                                               \
    "s_add_u32 s20, s20, Lt%=-Lh%=\n"                                                                       \
    "s_addc_u32 s21, s21, 0\n"                                                                              \
    "s_add_u32 s28, s20, " STR(N) "*128\n"                                                                  \
    "s_addc_u32 s29, s21, 0\n"                                                                              \
    "s_mov_b32 s22, %[start]\n"                                                                             \
    "s_and_b32 s22, s22, " STR(N) "-1\n"                                                                    \
    "s_mov_b32 s23, " STR(JUMPS) "\n"                                                                       \
    "s_lshl_b32 s24, s22, 7\n"                                                                              \
    "s_add_u32 s26, s20, s24\n"                                                                             \
    "s_addc_u32 s27, s21, 0\n"                                                                              \
    "s_setpc_b64 s[26:27]\n"                                                                                \
    ".p2align 7\n"                                                                                          \
    "Lt%=:\n"                                                                                               \
    ".rept " STR(N) "\n"                                                                                    \
    "v_fma_f64 %[a0], %[x], %[y], %[a0]\n v_fma_f64 %[a1], %[x], %[y], %[a1]\n v_fma_f64 %[a2], %[x], %[y], %[a2]\n" \
    "v_fma_f64 %[a3], %[x], %[y], %[a3]\n v_fma_f64 %[a0], %[x], %[y], %[a0]\n v_fma_f64 %[a1], %[x], %[y], %[a1]\n" \
    "s_mul_i32 s22, s22, 5\n"                                                                               \
    "s_add_u32 s22, s22, 1\n"                                                                               \
    "s_and_b32 s22, s22, " STR(N) "-1\n"                                                                    \
    "s_lshl_b32 s24, s22, 7\n"                                                                              \
    "s_add_u32 s26, s20, s24\n"                                                                             \
    "s_addc_u32 s27, s21, 0\n"                                                                              \
    "s_sub_u32 s23, s23, 1\n"                                                                               \
    "s_cmp_eq_u32 s23, 0\n"                                                                                 \
    "s_cselect_b32 s26, s28, s26\n"      /* the last jump goes behind the table */                          \
    "s_cselect_b32 s27, s29, s27\n"                                                                         \
    "s_setpc_b64 s[26:27]\n"                                                                                \
    ".p2align 7\n"                                                                                          \
    ".endr\n"                                                                                               \
    "Le%=:\n"

#define KERNEL(N)                                                                                           \
    __global__ void __launch_bounds__(256) k##N(double seed, double *out)                                   \
    {                                                                                                       \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + threadIdx.x, x = 1e-9, y = 0.5;         \
        const int start = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u >> 7)); \
        asm volatile(BODY(N)                                                                                \
                     : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)                           \
                     : [x] "v"(x), [y] "v"(y), [start] "s"(start)                                           \
                     : "s20", "s21", "s22", "s23", "s24", "s26", "s27", "s28", "s29", "scc");                             \
        if (a0 + a1 + a2 + a3 == 12345.678) out[0] = 1;                                                     \
    }
KERNEL(32)
KERNEL(64)
KERNEL(128)
KERNEL(256)
KERNEL(512)
KERNEL(1024)
KERNEL(2048)

// ---- lean blocks: 2 VALU + 8 scalar instructions, like the sweep kernel's step; tables up to 64 KB, aligned to 64 KiB so that the
// block address is base_lo + offset without a carry (as the sweep kernel forms it) ----------------------------------------------
#define LBODY(N)                                                                                            \
    "s_getpc_b64 s[20:21]\n"                                                                                \
    "Lh%=:\n"                                                                                               \
    "s_add_u32 s20, s20, Lt%=-Lh%=\n"                                                                       \
    "s_addc_u32 s21, s21, 0\n"                                                                              \
    "s_add_u32 s28, s20, " STR(N) "*128\n"                                                                  \
    "s_mov_b32 s27, s21\n"                                                                                  \
    "s_mov_b32 s22, %[start]\n"                                                                             \
    "s_and_b32 s22, s22, " STR(N) "-1\n"                                                                    \
    "s_mov_b32 s23, " STR(JUMPS) "\n"                                                                       \
    "s_lshl_b32 s24, s22, 7\n"                                                                              \
    "s_add_u32 s26, s20, s24\n"                                                                             \
    "s_setpc_b64 s[26:27]\n"                                                                                \
    ".p2align 16\n"                                                                                         \
    "Lt%=:\n"                                                                                               \
    ".rept " STR(N) "\n"                                                                                    \
    "v_fma_f64 %[a0], %[x], %[y], %[a0]\n v_fma_f64 %[a1], %[x], %[y], %[a1]\n"                             \
    "s_mul_i32 s22, s22, 5\n"                                                                               \
    "s_add_u32 s22, s22, 1\n"                                                                               \
    "s_and_b32 s22, s22, " STR(N) "-1\n"                                                                    \
    "s_lshl_b32 s24, s22, 7\n"                                                                              \
    "s_add_u32 s26, s20, s24\n"                                                                             \
    "s_sub_u32 s23, s23, 1\n"                                                                               \
    "s_cselect_b32 s26, s28, s26\n"      /* SCC = borrow: the jump after the last one goes behind the table */ \
    "s_setpc_b64 s[26:27]\n"                                                                                \
    ".p2align 7\n"                                                                                          \
    ".endr\n"

#define LKERNEL(N)                                                                                          \
    __global__ void __launch_bounds__(256) l##N(double seed, double *out)                                   \
    {                                                                                                       \
        double a0 = seed, a1 = seed + threadIdx.x, x = 1e-9, y = 0.5;                                        \
        const int start = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u >> 7)); \
        asm volatile(LBODY(N)                                                                               \
                     : [a0] "+v"(a0), [a1] "+v"(a1)                                                         \
                     : [x] "v"(x), [y] "v"(y), [start] "s"(start)                                           \
                     : "s20", "s21", "s22", "s23", "s24", "s26", "s27", "s28", "scc");                      \
        if (a0 + a1 == 12345.678) out[0] = 1;                                                               \
    }
LKERNEL(16)
LKERNEL(32)
LKERNEL(64)
LKERNEL(128)
LKERNEL(256)
LKERNEL(512)

template <typename F>
void run(F kern, int n, double *d)
{
    for (int wps : {1, 3, 4}) {   // waves per SIMD (blocks of 4 waves: one per SIMD of a CU)
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, 1.5, d);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, 1.5, d);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %4d = %6.1f KB  waves/SIMD %d: %.3f ms  %.1f ns per jump per wave  %.2f ns per jump per SIMD\n", n, n * 128 / 1024.0,
               wps, ms, ms * 1e6 / JUMPS, ms * 1e6 / JUMPS / wps);
    }
}

int main()
{
    double *d;
    (void)hipMalloc(&d, 1 << 16);
    printf("lean blocks (2 VALU + 8 scalar instructions):\n");
    run(l16, 16, d);
    run(l32, 32, d);
    run(l64, 64, d);
    run(l128, 128, d);
    run(l256, 256, d);
    run(l512, 512, d);
    printf("blocks of 6 VALU + 10 scalar instructions:\n");
    run(k32, 32, d);
    run(k64, 64, d);
    run(k128, 128, d);
    run(k256, 256, d);
    run(k512, 512, d);
    run(k1024, 1024, d);
    run(k2048, 2048, d);
    return 0;
}
