// Lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found empirically: A = 1 in ONE lane (la), B = 1 in ONE lane (lb), everything else 0 --
// which lane of D becomes 1?  Prints every (la, lb, ld) triple.  D[ld] sums four products (the contraction index), so every ld appears 4 times.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/mfma_f64_4x4_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) out[la * 64 + lb] = lane;
        }
}
int main() {
    int *d, h[4096];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0xFF, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb] >= 0) printf("%d %d %d\n", la, lb, h[la * 64 + lb]);
    return 0;
}
