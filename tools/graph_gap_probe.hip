// Is a hipGraph worth it for the launch-bound Arnoldi (complex n = 1e6: ~8 dependent kernels of 7-180 us per step)?
// Measures the time per kernel of a chain of N dependent kernels (a) launched on a stream, host running ahead,
// (b) captured once into a graph and replayed -- for an empty kernel and for a ~20 us streaming kernel.
//   hipcc --offload-arch=gfx950 -O3 -o graph_gap_probe graph_gap_probe.hip && ./graph_gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty(double *p, int i) { if (p && threadIdx.x == 9999) p[0] = i; }
__global__ void k_stream(double *p, long n, double a) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = p[i] * a + 1.0;
}
int main() {
    const int N = 1024;
    const long n = 4 << 20;   // 32 MB read + 32 MB write: ~15-20 us
    double *p; CK(hipMalloc(&p, n * sizeof(double))); CK(hipMemset(p, 0, n * sizeof(double)));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int which = 0; which < 2; ++which) {
        auto enqueue = [&]() {
            for (int i = 0; i < N; ++i) {
                if (which == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p, i);
                else hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, s, p, n, 1.0000001);
            }
        };
        enqueue(); CK(hipStreamSynchronize(s));
        double best_stream = 1e30, best_graph = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            enqueue(); CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best_stream) best_stream = us;
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        enqueue();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best_graph) best_graph = us;
        }
        printf("%s: %d dependent kernels  stream %.2f us/kernel   graph %.2f us/kernel\n", which == 0 ? "empty kernel " : "64 MB kernel ",
               N, best_stream / N, best_graph / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
