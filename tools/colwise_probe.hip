// Would the single-use sweeps (sweep 1: dots; sweep 3: update) stream faster ONE COLUMN AT A TIME?  A block owns a tile of
// R = 256 * 2 * U rows; y (or the running update u) lives in registers, and the block walks the k columns one after the other,
// reading R*8 contiguous bytes of each -- long contiguous runs per column instead of 1 KiB per column per wave.
//   hipcc --offload-arch=gfx950 -O3 -o colwise_probe colwise_probe.hip && ./colwise_probe [rows] [k]
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ double wsum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// dots: out[j] += sum_rows X(r, j) y(r)
template <int U>
__global__ __launch_bounds__(256) void dots_colwise(const double *__restrict__ X, long ld, int k, const double *__restrict__ y, long n,
                                                    double *__restrict__ out) {
    __shared__ double acc[256];                       // k <= 256 column sums of this block
    for (int j = threadIdx.x; j < k; j += 256) acc[j] = 0.0;
    __syncthreads();
    const long tile = 256L * 2 * U, ntiles = n / tile;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long r0 = t * tile + threadIdx.x * 2;
        v2d yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) yv[u] = *reinterpret_cast<const v2d *>(y + r0 + u * 512);
        for (int j = 0; j < k; ++j) {
            const double *xc = X + (long)j * ld + r0;
            v2d xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + u * 512));
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u) s = fma(xv[u].x, yv[u].x, fma(xv[u].y, yv[u].y, s));
            s = wsum(s);
            if ((threadIdx.x & 63) == 0) atomicAdd(&acc[j], s);      // LDS atomic (probe only: order not fixed)
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) out[(long)blockIdx.x * 256 + j] = acc[j];
}
// update: y <- y - X h (h in constant-ish global memory), one column at a time, u in registers
template <int U>
__global__ __launch_bounds__(256) void update_colwise(const double *__restrict__ X, long ld, int k, double *__restrict__ y, long n,
                                                      const double *__restrict__ h) {
    const long tile = 256L * 2 * U, ntiles = n / tile;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long r0 = t * tile + threadIdx.x * 2;
        v2d uacc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) uacc[u] = v2d{0.0, 0.0};
        for (int j = 0; j < k; ++j) {
            const double *xc = X + (long)j * ld + r0;
            const double hj = h[j];
#pragma unroll
            for (int u = 0; u < U; ++u) uacc[u] += __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + u * 512)) * hj;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v2d *yp = reinterpret_cast<v2d *>(y + r0 + u * 512);
            *yp = *yp - uacc[u];
        }
    }
}
template <typename F> float timeit(F f, hipStream_t s) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipStreamSynchronize(s);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { (void)hipEventRecord(a, s); f(); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
__global__ void fill_hash(double *p, size_t cnt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cnt; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}
// mode "scan": dots_colwise<4>, 3 blocks/CU, over a list of n (same allocation), zeros then pseudo-random data;
// mode "sustain": the same launch back to back for ~4 s, GB/s per group of launches (does a long run throttle?)
static int scan_main(int argc, char **argv) {
    const int k = argc > 3 ? atoi(argv[3]) : 128;
    const long nmax = argc > 4 ? atol(argv[4]) : 100000000;
    const bool sustain = !strcmp(argv[1], "sustain");
    double *X, *out; CK(hipMalloc(&X, (size_t)(nmax + 32) * (k + 1) * 8)); CK(hipMalloc(&out, 4096 * 256 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int data = 0; data < 2; ++data) {
        if (data == 0) CK(hipMemset(X, 0, (size_t)(nmax + 32) * (k + 1) * 8));
        else { hipLaunchKernelGGL(fill_hash, dim3(4096), dim3(256), 0, s, X, (size_t)(nmax + 32) * (k + 1)); CK(hipStreamSynchronize(s)); }
        if (sustain) {
            const long n = nmax, ld = n + 32;
            const double bd = 8.0 * n * (k + 1);
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            const int per = (int)(2.0e12 / bd) + 1;              // ~0.3 s of launches per group
            printf("sustain data=%s n=%ld k=%d, %d launches per group:", data ? "random" : "zeros", n, k, per);
            for (int g = 0; g < 14; ++g) {
                (void)hipEventRecord(a, s);
                for (int r = 0; r < per; ++r) hipLaunchKernelGGL((dots_colwise<4>), dim3(768), dim3(256), 0, s, X, ld, k, X + (long)k * ld, n, out);
                (void)hipEventRecord(b, s); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                printf(" %.0f", bd * per / ms / 1e6);
            }
            printf("\n");
            continue;
        }
        for (long n : {1000000L, 2000000L, 4000000L, 8000000L, 16000000L, 32000000L, 64000000L, 100000000L}) {
            if (n > nmax) break;
            const long ld = n + 32;
            const double bd = 8.0 * n * (k + 1);
            float ms = timeit([&] { hipLaunchKernelGGL((dots_colwise<4>), dim3(768), dim3(256), 0, s, X, ld, k, X + (long)k * ld, n, out); }, s);
            printf("scan data=%s n=%ld k=%d  %.0f GB/s (best of 5, %.3f ms)\n", data ? "random" : "zeros", n, k, bd / ms / 1e6, ms);
        }
    }
    return 0;
}
int main(int argc, char **argv) {
    if (argc > 1 && (!strcmp(argv[1], "scan") || !strcmp(argv[1], "sustain"))) return scan_main(argc, argv);
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    const int k = argc > 2 ? atoi(argv[2]) : 128;
    const long ld = n + 32;
    double *X, *out, *h; CK(hipMalloc(&X, (size_t)ld * (k + 1) * 8)); CK(hipMalloc(&out, 4096 * 256 * 8)); CK(hipMalloc(&h, 256 * 8));
    CK(hipMemset(X, 0, (size_t)ld * (k + 1) * 8)); CK(hipMemset(h, 0, 256 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("n = %ld, k = %d   kernel U(16-B loads per lane per column) blocks/CU  GB/s\n", n, k);
#define RUN(K, U, BYTES, ...) for (int m : {1, 2, 3, 4, 6, 8}) { float ms = timeit([&] { hipLaunchKernelGGL((K<U>), dim3(256 * m), dim3(256), 0, s, __VA_ARGS__); }, s); \
        printf(#K " %d %d  %.0f\n", U, m, BYTES / ms / 1e6); }
    const double bd = 8.0 * n * (k + 1), bu = 8.0 * n * (k + 2);
    RUN(dots_colwise, 1, bd, X, ld, k, X + (long)k * ld, n, out) RUN(dots_colwise, 2, bd, X, ld, k, X + (long)k * ld, n, out)
    RUN(dots_colwise, 4, bd, X, ld, k, X + (long)k * ld, n, out) RUN(dots_colwise, 8, bd, X, ld, k, X + (long)k * ld, n, out)
    RUN(update_colwise, 1, bu, X, ld, k, X + (long)k * ld, n, h) RUN(update_colwise, 2, bu, X, ld, k, X + (long)k * ld, n, h)
    RUN(update_colwise, 4, bu, X, ld, k, X + (long)k * ld, n, h) RUN(update_colwise, 8, bu, X, ld, k, X + (long)k * ld, n, h)
    return 0;
}
