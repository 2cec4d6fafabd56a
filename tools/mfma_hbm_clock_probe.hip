// Do FP64 MFMAs and an HBM stream slow each other down on the SAME chip at the same time, and is it the clock?  (round 5)
// Every CU runs two blocks of four waves: "m" blocks issue back-to-back v_mfma_f64_16x16x4_f64 (VGPR form, inline assembly: the ceiling of
// tools/mfma_f64_peak.hip), "s" blocks stream a large buffer with 16-byte non-temporal loads (a sum keeps them alive).  Three runs: MFMA blocks
// alone, stream blocks alone, both.  Each MFMA wave reads s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its loop: their
// ratio is the average shader clock while it ran.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_hbm_clock_probe tools/mfma_hbm_clock_probe.hip && ./mfma_hbm_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void probe(int mode, int iters, int passes, const v2d *__restrict__ buf, int64_t nvec, double *out, unsigned long long *clk,
                                             unsigned long long *sclk) {
    // even blocks: MFMA (mode & 1), odd blocks: stream (mode & 2)
    const bool is_m = (blockIdx.x & 1) == 0;
    if (is_m && (mode & 4)) {
        // LDS-fed MFMAs: ten 8-byte operand reads per nine MFMAs (the ratio of the library's Gram kernel), reads one group ahead, pinned
        __shared__ double tile[64 * 34];
        for (int i = threadIdx.x; i < 64 * 34; i += 256) tile[i] = 1.0 + 1e-3 * (i & 63);
        __syncthreads();
        const int lane = threadIdx.x & 63, o = (lane & 15) * 34 + (lane >> 4);
        v4d acc[5];
        for (int i = 0; i < 5; ++i) acc[i] = v4d{0, 0, 0, 0};
        double rn[10];
        auto fetch = [&](int it) {
#pragma unroll
            for (int q = 0; q < 10; ++q) rn[q] = tile[o + 16 * 34 * (q & 3) + 4 * ((it + q) & 7)];
        };
        fetch(0);
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters * 7; ++it) {               // 9 MFMAs per iteration: iters * 63 in all (against iters * 64 of the register-fed loop)
            double r[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) r[q] = rn[q];
            fetch(it + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int d = 0; d < 4; ++d) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[d]) : "v"(r[4 * h]), "v"(r[4 * h + d]));
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[4]) : "v"(r[8]), "v"(r[9]));
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        double s = 0;
        for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (threadIdx.x == 0) { clk[2 * (blockIdx.x >> 1)] = c1 - c0; clk[2 * (blockIdx.x >> 1) + 1] = r1 - r0; }
    } else if (is_m) {
        if (!(mode & 1)) return;
        v4d acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
        const double av = 1.0000001 + 1e-3 * (threadIdx.x & 15), bv = 0.9999999 - 1e-3 * (threadIdx.x >> 4);
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        double s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (threadIdx.x == 0) { clk[2 * (blockIdx.x >> 1)] = c1 - c0; clk[2 * (blockIdx.x >> 1) + 1] = r1 - r0; }
    } else {
        if (!(mode & 2)) return;
        const int64_t nblk = gridDim.x >> 1, b = blockIdx.x >> 1;
        v2d s = v2d{0, 0};
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
        for (int ps = 0; ps < passes; ++ps)
        for (int64_t i = b * 256 + threadIdx.x; i < nvec; i += nblk * 256 * 4) {
            const v2d a0 = __builtin_nontemporal_load(buf + i);
            const int64_t i1 = i + nblk * 256, i2 = i + 2 * nblk * 256, i3 = i + 3 * nblk * 256;
            const v2d a1 = i1 < nvec ? __builtin_nontemporal_load(buf + i1) : v2d{0, 0};
            const v2d a2 = i2 < nvec ? __builtin_nontemporal_load(buf + i2) : v2d{0, 0};
            const v2d a3 = i3 < nvec ? __builtin_nontemporal_load(buf + i3) : v2d{0, 0};
            s += (a0 + a1) + (a2 + a3);
        }
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
        if (threadIdx.x == 0) sclk[b] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
int main() {
    const int64_t bytes = (int64_t)12 << 30, nvec = bytes / 16;
    v2d *buf; double *out; unsigned long long *clk;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    const int grid = 256 * 2 * 4;                   // per CU: four MFMA blocks and four stream blocks of four waves
    (void)hipMalloc(&out, sizeof(double) * grid * 256);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * grid);
    unsigned long long *sclk; (void)hipMalloc(&sclk, sizeof(unsigned long long) * grid);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 3600, passes = 11;            // 64 MFMAs per iteration: ~28 ms of MFMAs; 11 passes over 12 GB: ~27 ms of streaming
    for (int rep = 0; rep < 2; ++rep)
        for (int mode : {1, 2, 3, 5, 7}) {
            (void)hipMemset(clk, 0, sizeof(unsigned long long) * grid);
            (void)hipMemset(sclk, 0, sizeof(unsigned long long) * grid);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, mode, iters, passes, buf, nvec, out, clk, sclk);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid);
            (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
            double cs = 0, rs = 0; int cnt = 0;
            for (int b = 0; b < grid / 2; ++b) if (h[2 * b + 1]) { cs += (double)h[2 * b]; rs += (double)h[2 * b + 1]; ++cnt; }
            const double mfma_us = cnt ? rs / cnt / 100.0 : 0.0;          // 100 MHz reference
            std::vector<unsigned long long> hs(grid);
            (void)hipMemcpy(hs.data(), sclk, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
            double smax = 0; for (int b = 0; b < grid / 2; ++b) if ((double)hs[b] > smax) smax = (double)hs[b];
            const double stream_us = smax / 100.0;
            const double tf = (mode & 1) ? (double)(grid / 2) * 4 * iters * ((mode & 4) ? 63.0 : 64.0) * 2048.0 / (mfma_us * 1e-6) / 1e12 : 0.0;
            printf("{\"mode\": \"%s\", \"kernel_ms\": %.3f, \"MFMA_TFLOPs_while_the_MFMA_waves_ran\": %.1f, \"shader_clock_GHz_of_the_MFMA_waves\": %.3f, \"stream_GBps_while_the_stream_blocks_ran\": %.0f, \"MFMA_ms\": %.2f, \"stream_ms\": %.2f}\n",
                   mode == 1 ? "MFMA alone" : (mode == 2 ? "stream alone" : (mode == 3 ? "MFMA + stream" : (mode == 5 ? "LDS-fed MFMA alone" : "LDS-fed MFMA + stream"))), ms, tf, cnt ? cs / rs * 0.1 : 0.0, (mode & 2) ? (double)passes * bytes / (stream_us * 1e-6) / 1e9 : 0.0, mfma_us * 1e-3, stream_us * 1e-3);
        }
    return 0;
}
