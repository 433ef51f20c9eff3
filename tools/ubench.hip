// ubench.hip — instruction-cost probes for gfx950 used to steer the INSIDER kernels (diagnostic, not product).
// Each probe runs REPS iterations of an unrolled block of UNR identical instructions per wave and reports
// cycles per instruction per wave (s_memtime) for `waves_per_simd` resident waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef double d4 __attribute__((ext_vector_type(4)));
__device__ inline d4 dz() { d4 z; z[0] = 0; z[1] = 0; z[2] = 0; z[3] = 0; return z; }
#define REPS 2000

#define PROBE(NAME, SETUP, BODY16, SINK)                                                        \
    __global__ void NAME(unsigned long long *out, double seed)                                  \
    {                                                                                           \
        SETUP;                                                                                  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int r = 0; r < REPS; ++r) { BODY16; }                                              \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        SINK;                                                                                   \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

#define R4(X) X X X X
#define R16(X) R4(R4(X))

// dependent chains
PROBE(k_fma_dep, double a = seed + threadIdx.x; double b = 1.0000001; double c = 1e-9,
      R16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));), if (a == 12345.678) out[0] = 1)
PROBE(k_add_dep, double a = seed + threadIdx.x; double b = 1e-9,
      R16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));), if (a == 12345.678) out[0] = 1)
PROBE(k_mul_dep, double a = seed + threadIdx.x; double b = 1.0000001,
      R16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));), if (a == 12345.678) out[0] = 1)
PROBE(k_max_dep, double a = seed + threadIdx.x; double b = 0.5,
      R16(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));), if (a == 12345.678) out[0] = 1)
PROBE(k_add32_dep, int a = (int)seed + threadIdx.x; int b = 3,
      R16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));), if (a == 12345678) out[0] = 1)
// independent streams (4 accumulators)
PROBE(k_fma_ind, double a0 = seed; double a1 = seed + 1; double a2 = seed + 2; double a3 = seed + 3; double b = 1.0000001; double c = 1e-9,
      R4(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));),
      if (a0 + a1 + a2 + a3 == 12345.678) out[0] = 1)
PROBE(k_add32_ind, int a0 = 1; int a1 = 2; int a2 = 3; int a3 = 4; int b = (int)seed,
      R4(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));),
      if (a0 + a1 + a2 + a3 == 12345678) out[0] = 1)
PROBE(k_salu_dep, int a = 0,
      R16(asm volatile("s_add_u32 s40, s40, 3" ::: "s40");), (void)a)
PROBE(k_readlane_fma, double a = seed + threadIdx.x; double g = 1e-9; int k = __builtin_amdgcn_readfirstlane(((int)seed) & 31),
      R16({ int lo = __builtin_amdgcn_readlane(__double2loint(a), k); int hi = __builtin_amdgcn_readlane(__double2hiint(a), k);
            double d = __hiloint2double(hi, lo); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(d), "v"(g)); }),
      if (a == 12345.678) out[0] = 1)
PROBE(k_bperm_fma, double a = seed + threadIdx.x; double g = 1e-9; int addr = ((threadIdx.x & 32) | 5) << 2,
      R16({ int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(a)); int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(a));
            double d = __hiloint2double(hi, lo); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(d), "v"(g)); }),
      if (a == 12345.678) out[0] = 1)
// f64 MFMA: 4 independent accumulators, and a single dependent accumulator
PROBE(k_mfma_ind, d4 c0 = dz(); d4 c1 = c0; d4 c2 = c0; d4 c3 = c0; double a = seed + threadIdx.x; double b = 1.0 + threadIdx.x,
      R4({ c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
           c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0); }),
      if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) out[0] = 1)
PROBE(k_mfma_dep, d4 c0 = dz(); double a = seed + threadIdx.x; double b = 1.0 + threadIdx.x,
      R16({ c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); }),
      if (c0[0] == 12345.678) out[0] = 1)
// MFMA with 4 independent fp64 FMAs between MFMAs (does the VALU work hide under the matrix pipe?)
PROBE(k_mfma_valu, d4 c0 = dz(); d4 c1 = c0; double a = seed + threadIdx.x; double b = 1.0 + threadIdx.x;
      double a0 = seed; double a1 = seed + 1; double a2 = seed + 2; double a3 = seed + 3; double bb = 1.0000001; double cc = 1e-9,
      R4({ c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
           asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(bb), "v"(cc));
           c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
           asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(bb), "v"(cc)); }),
      if (c0[0] + c1[1] + a0 + a1 + a2 + a3 == 12345.678) out[0] = 1)

struct P { const char *name; void (*fn)(unsigned long long *, double); int per_iter; };

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 1 << 20);
    P probes[] = {{"v_fma_f64 dependent", k_fma_dep, 16},
                  {"v_fma_f64 independent x4", k_fma_ind, 16},
                  {"readlane x2 + v_fma_f64 (dependent chain)", k_readlane_fma, 16},
                  {"ds_bpermute x2 + v_fma_f64 (dependent chain)", k_bperm_fma, 16},
                  {"mfma_f64_16x16x4 independent x4", k_mfma_ind, 16}, {"mfma_f64_16x16x4 dependent", k_mfma_dep, 16},
                  {"mfma_f64 + 4 v_fma_f64 per mfma (per mfma)", k_mfma_valu, 8}};
    for (auto &p : probes)
        for (int wps : {1, 2, 4}) {
            const int threads = 64 * 4 * wps;          // one block per CU: 4 SIMDs x wps waves
            const int blocks = 256;
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(threads > 1024 ? 1024 : threads), 0, 0, d, 1.5);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    std::vector<unsigned long long> h(blocks * (threads / 64));
                    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
                    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
                    // s_memtime counts at 100 MHz: convert with wall time instead -> report ns per instr per wave and per SIMD
                    double n_inst = (double)REPS * p.per_iter;
                    printf("%-48s waves/SIMD=%d  %7.2f ns/instr/wave  %7.2f ns/instr/SIMD (%.1f cyc @2.4GHz)  [memtime ticks/instr %.2f]\n",
                           p.name, wps > 4 ? 4 : wps, ms * 1e6 / n_inst, ms * 1e6 / n_inst / (wps > 4 ? 4 : wps),
                           ms * 1e6 / n_inst / (wps > 4 ? 4 : wps) * 2.4, avg / n_inst);
                }
            }
        }
    return 0;
}
