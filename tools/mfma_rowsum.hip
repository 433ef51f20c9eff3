// Can a chain of two v_mfma_f64_4x4x4 add up the 16 lanes of each DPP row (the per-sweep loss change of the sweep kernel)?
// Result on MI355X: NO, for all four operand orders: the contraction index is lane / 16 (the row number) and the four blocks
// are 4-lane groups inside a row, so the products sum ACROSS rows.  The sweep kernel keeps its 4-step DPP reduction.
// hipcc --offload-arch=gfx950 -O2 -o mfma_rowsum mfma_rowsum.hip && ./mfma_rowsum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *out)
{
    const int l = threadIdx.x;
    const double v = x[l], one = 1.0;
    double r;
    r = __builtin_amdgcn_mfma_f64_4x4x4f64(v, one, 0.0, 0, 0, 0);
    out[0 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(r, one, 0.0, 0, 0, 0);
    out[1 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(one, r, 0.0, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f64_4x4x4f64(one, v, 0.0, 0, 0, 0);
    out[2 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(r, one, 0.0, 0, 0, 0);
    out[3 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(one, r, 0.0, 0, 0, 0);
}
int main()
{
    double hx[64], ho[256], *dx, *dout;
    for (int i = 0; i < 64; ++i) hx[i] = std::ldexp(1.0, i % 16) + 65536.0 * (i / 16);   // distinct bits per lane
    hipMalloc(&dx, sizeof hx); hipMalloc(&dout, sizeof ho);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    for (int c = 0; c < 4; ++c) {
        int good = 0;
        for (int l = 0; l < 64; ++l) {
            double s = 0; for (int i = 0; i < 16; ++i) s += hx[(l & 48) + i];
            good += ho[c * 64 + l] == s;
        }
        printf("chain %d: %d of 64 lanes hold their row's sum (lane 0: %.1f, lane 17: %.1f)\n", c, good, ho[c * 64], ho[c * 64 + 17]);
    }
    return 0;
}
