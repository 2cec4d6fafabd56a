// Streaming-kernel shapes for the BLAS-1 primitives (scal: 1 read + 1 write in place; axpby: 2 reads + 1 write; dot: 2 reads),
// n = 1e8 doubles: loads per lane in flight (U), non-temporal hints, blocks per CU.  Picks the shape lk_kernels.hip.h uses.
//   hipcc --offload-arch=gfx950 -O3 -o blas1_probe blas1_probe.hip && ./blas1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void scal(v2d *__restrict__ x, long nv, double a) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nv; i += U * stride) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] *= a; if (NT) __builtin_nontemporal_store(v[u], x + i + u * stride); else x[i + u * stride] = v[u]; }
    }
    for (; i < nv; i += stride) x[i] = x[i] * a;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void axpby(const v2d *__restrict__ x, v2d *__restrict__ y, long nv, double a, double b) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < nv; i += U * stride) {
        v2d v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride]; w[u] = NT ? __builtin_nontemporal_load(y + i + u * stride) : y[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { v2d r = v[u] * a + w[u] * b; if (NT) __builtin_nontemporal_store(r, y + i + u * stride); else y[i + u * stride] = r; }
    }
    for (; i < nv; i += stride) y[i] = x[i] * a + y[i] * b;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void dot(const v2d *__restrict__ x, const v2d *__restrict__ y, long nv, double *out) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    v2d acc = {0, 0};
    for (; i + (U - 1) * stride < nv; i += U * stride) {
        v2d v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride]; w[u] = NT ? __builtin_nontemporal_load(y + i + u * stride) : y[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u] * w[u];
    }
    for (; i < nv; i += stride) acc += x[i] * y[i];
    if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}
template <typename F> float timeit(F f, hipStream_t s) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipStreamSynchronize(s);
    float best = 1e30f;
    for (int r = 0; r < 7; ++r) { hipEventRecord(a, s); f(); hipEventRecord(b, s); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    const long n = 100000000, nv = n / 2;
    v2d *x, *y; double *out;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int mults[] = {1, 2, 3, 4, 6, 8};
    printf("kernel U nt blocks/CU  GB/s\n");
#define RUN(K, U, NT, BYTES, ...) for (int m : mults) { int g = 256 * m; float ms = timeit([&] { hipLaunchKernelGGL((K<U, NT>), dim3(g), dim3(256), 0, s, __VA_ARGS__); }, s); printf(#K " %d %d %2d  %.0f\n", U, (int)NT, m, BYTES / ms / 1e6); }
#define ALL(K, BYTES, ...) RUN(K, 1, false, BYTES, __VA_ARGS__) RUN(K, 1, true, BYTES, __VA_ARGS__) RUN(K, 2, true, BYTES, __VA_ARGS__) RUN(K, 4, true, BYTES, __VA_ARGS__)
    ALL(scal, 2.0 * n * 8, x, nv, 1.0000001)
    ALL(axpby, 3.0 * n * 8, x, y, nv, 0.5, 0.999)
    ALL(dot, 2.0 * n * 8, x, y, nv, out)
    return 0;
}
