// How many loads should a dot sweep keep in flight?  The BLAS-1 probe (blas1_probe.hip) peaks at 16-32 KiB in flight per
// CU (7.2 TB/s for a two-stream dot) and LOSES bandwidth beyond; lk::panel_sweep<DOT> keeps 8 waves x 17 KiB = 136 KiB.
// This is the dot sweep's skeleton (lanes along rows, WC waves across columns, KC columns per wave, accumulators in
// registers, grid-stride tiles) with the loads of a wave issued in batches of B columns.
//   hipcc --offload-arch=gfx950 -O3 -o sweep_probe sweep_probe.hip && ./sweep_probe [rows] [k]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NW, int KC, int B>
__global__ __launch_bounds__(NW * 64) void sweep(const double *__restrict__ X, long ld, int k, const double *__restrict__ y, long n,
                                                 double *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int WC = (k + KC - 1) / KC, WR = NW / WC;
    const int wc = wave % WC, wr = wave / WC;
    const long tile_rows = (long)WR * 128, ntiles = n / tile_rows;
    const double *Xw = X + (long)wc * KC * ld;
    v2d acc[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) acc[j] = v2d{0, 0};
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long r = t * tile_rows + wr * 128 + lane * 2;
        const v2d yv = *reinterpret_cast<const v2d *>(y + r);
#pragma unroll 1
        for (int jb = 0; jb < KC; jb += B) {
            v2d xv[B];
#pragma unroll
            for (int j = 0; j < B; ++j) xv[j] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(Xw + (long)(jb + j) * ld + r));
            // acc[jb + j] with a runtime jb would spill: rotate the accumulators instead (KC/B rotations per tile = identity)
#pragma unroll
            for (int j = 0; j < B; ++j) acc[j] += xv[j] * yv;
            if (B < KC) {
                v2d tmp[B];
#pragma unroll
                for (int j = 0; j < B; ++j) tmp[j] = acc[j];
#pragma unroll
                for (int j = 0; j + B < KC; ++j) acc[j] = acc[j + B];
#pragma unroll
                for (int j = 0; j < B; ++j) acc[KC - B + j] = tmp[j];
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < KC; ++j) s += acc[j].x + acc[j].y;
    if (s == 1.2345e300) out[0] = s;
}
// The same skeleton with U row segments per wave: a wave holds KC columns x U x 128 rows of the tile in registers (what an
// update + dot sweep needs), so each column contributes U KiB... of U separate 1-KiB wave loads 1 KiB apart?  No: the U
// segments of a wave are CONTIGUOUS (rows [w*U*128, (w+1)*U*128)), i.e. U KiB of consecutive rows per column and wave.
template <int NW, int KC, int U>
__global__ __launch_bounds__(NW * 64) void sweepU(const double *__restrict__ X, long ld, int k, const double *__restrict__ y, long n,
                                                  double *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int WC = (k + KC - 1) / KC, WR = NW / WC;
    const int wc = wave % WC, wr = wave / WC;
    const long tile_rows = (long)WR * 128 * U, ntiles = n / tile_rows;
    const double *Xw = X + (long)wc * KC * ld;
    v2d acc[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) acc[j] = v2d{0, 0};
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long r = t * tile_rows + (long)wr * 128 * U + lane * 2;
        v2d yv[U], xv[KC][U];
#pragma unroll
        for (int u = 0; u < U; ++u) yv[u] = *reinterpret_cast<const v2d *>(y + r + u * 128);
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u) xv[j][u] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(Xw + (long)j * ld + r + u * 128));
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int u = 0; u < U; ++u) acc[j] += xv[j][u] * yv[u];
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < KC; ++j) s += acc[j].x + acc[j].y;
    if (s == 1.2345e300) out[0] = s;
}
template <typename F> float timeit(F f, hipStream_t s) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipStreamSynchronize(s);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { (void)hipEventRecord(a, s); f(); (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    const int k = argc > 2 ? atoi(argv[2]) : 128;
    const long ld = n + 32;
    double *X, *out; CK(hipMalloc(&X, (size_t)ld * (k + 1) * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemset(X, 0, (size_t)ld * (k + 1) * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    const double bytes = 8.0 * n * (k + 1);
    printf("n = %ld, k = %d: NW KC batch blocks/CU  GB/s\n", n, k);
#define RUN(NW, KC, B) for (int m : {1, 2, 3, 4}) { float ms = timeit([&] { hipLaunchKernelGGL((sweep<NW, KC, B>), dim3(256 * m), dim3(NW * 64), 0, s, X, ld, k, X + (long)k * ld, n, out); }, s); \
        printf("%2d %2d %2d %d  %.0f\n", NW, KC, B, m, bytes / ms / 1e6); }
    if (argc > 3) {
        printf("row segments: NW KC U blocks/CU  GB/s\n");
#define RUNU(NW, KC, U) for (int m : {1, 2, 3, 4}) { float ms = timeit([&] { hipLaunchKernelGGL((sweepU<NW, KC, U>), dim3(256 * m), dim3(NW * 64), 0, s, X, ld, k, X + (long)k * ld, n, out); }, s); \
        printf("U %2d %2d %2d %d  %.0f\n", NW, KC, U, m, bytes / ms / 1e6); }
        RUNU(8, 16, 1) RUNU(8, 16, 2) RUNU(16, 8, 1) RUNU(16, 8, 2) RUNU(16, 8, 4) RUNU(8, 16, 4)
        return 0;
    }
    RUN(8, 16, 16) RUN(8, 16, 8) RUN(8, 16, 4) RUN(8, 16, 2)
    RUN(4, 32, 8) RUN(4, 32, 4) RUN(4, 32, 2)
    RUN(16, 8, 8) RUN(16, 8, 4) RUN(16, 8, 2)
    return 0;
}
