// FP64 MFMA issue-rate microbenchmark (calibration of the tall-skinny product's ceiling): every wave of a full grid issues
// back-to-back v_mfma_f64_16x16x4_f64 on 8 independent accumulators; prints TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak tools/mfma_f64_peak.hip && ./mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kfma(double *out, int iters, double a, double b) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *d;
    const int blocks = 256 * 8, iters = 20000;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 0.9999999);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 /*waves*/ * iters * 8.0 * 2048.0;
        printf("mfma_f64_16x16x4: %.3f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double flop2 = (double)blocks * 256 * iters * 16.0 * 2.0;
        printf("v_fma_f64       : %.3f ms  %.1f TFLOP/s\n", ms, flop2 / ms / 1e9);
    }
    return 0;
}
