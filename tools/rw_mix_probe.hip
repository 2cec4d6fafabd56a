// Probe of the HBM read/write mix behind DGS sweep 3 (DESIGN.md tuning log): 512 persistent blocks stream-read K
// "columns" of n doubles (16 B per lane, non-temporal) and write ONE column of results,
//   mode 0: no write (read-only reference)
//   mode 1: trickle -- every tile's 1 KiB is stored as soon as it is computed (what panel_sweep does)
//   mode 2: phases  -- results parked in LDS; every M tiles the whole grid synchronises and all blocks flush together
//   mode 3: cyclic tiles, results parked in LDS, each block flushes its M parked tiles on its own (no grid sync)
//   mode 4/5/6: BLOCKED tile mapping (a block owns a contiguous run of tiles): 4 = read-only, 5 = trickle, 6 = each block
//               flushes M consecutive tiles = M KiB contiguous
//   hipcc --offload-arch=gfx950 -O3 -o rw_mix_probe tools/rw_mix_probe.hip && ./rw_mix_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int K = 32;            // columns read per row
constexpr int NW = 8;            // waves per block; each wave owns K/NW columns of the tile, like the sweep's column split
constexpr int M = 32;            // tiles parked per block between flushes (32 KiB of LDS)

template <int MODE>
__global__ __launch_bounds__(NW * 64) void probe(const double *__restrict__ X, int64_t ld, double *__restrict__ y, int64_t n) {
    cg::grid_group grid = cg::this_grid();
    __shared__ v2d part[NW * 64];
    constexpr bool PARK = MODE == 2 || MODE == 3 || MODE == 6;
    constexpr bool BLOCKED = MODE >= 4;
    __shared__ v2d park[PARK ? M * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile_rows = 128, ntiles = n / tile_rows;
    const int64_t per_block = (ntiles + gridDim.x - 1) / gridDim.x;   // same trip count for every block (grid.sync inside)
    for (int64_t it = 0; it < per_block; ++it) {
        const int64_t t = BLOCKED ? (int64_t)blockIdx.x * per_block + it : blockIdx.x + it * gridDim.x;
        const bool live = t < ntiles;
        v2d s = v2d{0.0, 0.0};
        if (live) {
            const int64_t r = t * tile_rows + lane * 2;
#pragma unroll
            for (int j = 0; j < K / NW; ++j)
                s += __builtin_nontemporal_load(reinterpret_cast<const v2d *>(X + (int64_t)(wave * (K / NW) + j) * ld + r));
        }
        part[wave * 64 + lane] = s;
        __syncthreads();
        if (wave == 0) {
            v2d tot = v2d{0.0, 0.0};
            for (int w = 0; w < NW; ++w) tot += part[w * 64 + lane];
            if ((MODE == 1 || MODE == 5) && live) *reinterpret_cast<v2d *>(y + t * tile_rows + lane * 2) = tot;
            if (PARK) park[(it % M) * 64 + lane] = tot;
        }
        __syncthreads();
        if (PARK && ((it % M) == M - 1 || it == per_block - 1)) {
            if (MODE == 2) grid.sync();                   // reads pause chip-wide ...
            const int cnt = (int)(it % M) + 1;
            for (int sidx = wave; sidx < cnt; sidx += NW) {
                const int64_t itx = it - (cnt - 1) + sidx;
                const int64_t ts = BLOCKED ? (int64_t)blockIdx.x * per_block + itx : blockIdx.x + itx * gridDim.x;
                if (ts < ntiles) *reinterpret_cast<v2d *>(y + ts * tile_rows + lane * 2) = park[sidx * 64 + lane];
            }
            if (MODE == 2) grid.sync();                   // ... until every block has flushed
            else __syncthreads();
        }
    }
}

// mode 7: the SAME bytes written by 8 dedicated blocks in 256-KiB contiguous chunks while the other 504 only read
// (what a producer -> writer funnel through L2 would look like from the HBM's side; the written data is fake)
__global__ __launch_bounds__(NW * 64) void probe_funnel(const double *__restrict__ X, int64_t ld, double *__restrict__ y, int64_t n) {
    __shared__ v2d part[NW * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WRITERS = 8;
    if (blockIdx.x < WRITERS) {
        const int64_t nv = n / 2, per = (nv + WRITERS - 1) / WRITERS;
        v2d *yv = reinterpret_cast<v2d *>(y);
        const int64_t lo = blockIdx.x * per, hi = (lo + per < nv) ? lo + per : nv;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) yv[i] = v2d{1.0, 2.0};
        return;
    }
    const int64_t tile_rows = 128, ntiles = n / tile_rows;
    const int readers = gridDim.x - WRITERS;
    for (int64_t t = blockIdx.x - WRITERS; t < ntiles; t += readers) {
        const int64_t r = t * tile_rows + lane * 2;
        v2d s = v2d{0.0, 0.0};
#pragma unroll
        for (int j = 0; j < K / NW; ++j)
            s += __builtin_nontemporal_load(reinterpret_cast<const v2d *>(X + (int64_t)(wave * (K / NW) + j) * ld + r));
        part[wave * 64 + lane] = s;
        __syncthreads();
        if (wave == 0 && s.x == 12345.678) y[0] = part[lane].x;      // keep the loads alive
        __syncthreads();
    }
}

template <int MODE>
float run(const double *X, int64_t ld, double *y, int64_t n, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    void *args[] = {(void *)&X, (void *)&ld, (void *)&y, (void *)&n};
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (hipLaunchCooperativeKernel((const void *)probe<MODE>, dim3(blocks), dim3(NW * 64), args, 0, 0) != hipSuccess) {
            printf("cooperative launch refused for mode %d\n", MODE);
            return -1.f;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const int64_t n = 50000000;                    // 0.4 GB per column, 12.8 GB read per run
    double *X, *y;
    if (hipMalloc(&X, sizeof(double) * n * K) != hipSuccess || hipMalloc(&y, sizeof(double) * n) != hipSuccess) return 1;
    hipMemset(X, 0, sizeof(double) * n * K);
    const int blocks = 512;
    const double gb = 8.0 * n * K / 1e9, wgb = 8.0 * n / 1e9;
    const float t0 = run<0>(X, n, y, n, blocks), t1 = run<1>(X, n, y, n, blocks), t2 = run<2>(X, n, y, n, blocks);
    const float t3 = run<3>(X, n, y, n, blocks), t4 = run<4>(X, n, y, n, blocks), t5 = run<5>(X, n, y, n, blocks),
                t6 = run<6>(X, n, y, n, blocks);
    float t7 = 1e30f;
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe_funnel, dim3(blocks), dim3(NW * 64), 0, 0, X, n, y, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < t7) t7 = ms;
        }
    }
    printf("K=%d columns, n=%lld rows, %d blocks: read %.1f GB, write %.2f GB\n", K, (long long)n, blocks, gb, wgb);
    printf("mode 0 read-only : %7.3f ms  %6.0f GB/s\n", t0, gb / t0 * 1e3);
    printf("mode 1 trickle   : %7.3f ms  %6.0f GB/s (reads+writes)   store cost %.3f ms\n", t1, (gb + wgb) / t1 * 1e3, t1 - t0);
    printf("mode 2 phases    : %7.3f ms  %6.0f GB/s (reads+writes)   store cost %.3f ms\n", t2, (gb + wgb) / t2 * 1e3, t2 - t0);
    printf("mode 3 park/cycl : %7.3f ms  store cost %.3f ms\n", t3, t3 - t0);
    printf("mode 4 blocked RO: %7.3f ms  %6.0f GB/s\n", t4, gb / t4 * 1e3);
    printf("mode 5 blocked tr: %7.3f ms  store cost %.3f ms (vs blocked read-only)\n", t5, t5 - t4);
    printf("mode 6 blocked pk: %7.3f ms  store cost %.3f ms (vs blocked read-only)\n", t6, t6 - t4);
    printf("mode 7 funnel    : %7.3f ms  store cost %.3f ms (8 writer blocks, contiguous)\n", t7, t7 - t0);
    return 0;
}
