// FP64 MFMA issue-rate microbenchmark (calibration of the ceiling of every matrix-core kernel of this library): every wave of a full grid
// issues back-to-back v_mfma_f64_16x16x4_f64 on NACC independent accumulators, nothing else.  Prints, per configuration (waves per SIMD,
// accumulators per wave): TFLOP/s by wall clock, shader cycles per MFMA and SIMD by s_memtime, and the shader clock the two imply.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak tools/mfma_f64_peak.hip && ./mfma_f64_peak
// Spec (AMD): 78.6 TFLOP/s FP64 matrix = 256 CUs x 4 SIMDs x 2.4 GHz x 2048 flop / 64 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *cyc, int iters, double a, double b) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    const double av = a + 1e-3 * (threadIdx.x & 15), bv = b - 1e-3 * (threadIdx.x >> 4);      // operands that differ between lanes
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the k-step of the library's tall-skinny product without its loads: 16 accumulators, acc[4 g + h] += A[g] B[h] (every A operand feeds four
// consecutive MFMAs), STEPS such steps per loop iteration -- what the 16x16x4 instruction sustains in that shape, loop overhead amortised
template <int STEPS>
__global__ __launch_bounds__(256) void kpat(double *out, int iters, double a, double b) {
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4d{0, 0, 0, 0};
    double A[4], B[4];
    for (int i = 0; i < 4; ++i) { A[i] = a + 1e-3 * (threadIdx.x & 15) + 0.01 * i; B[i] = b - 1e-3 * (threadIdx.x >> 4) - 0.01 * i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[4 * g + h] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[g], B[h], acc[4 * g + h], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// The instruction itself, in its VGPR form by inline assembly (no compiler choice of accumulator registers, no copies): NACC independent accumulators,
// 64 / NACC rounds per loop iteration.  THIS is the issue-rate ceiling; k<> above is kept because rounds 4-5 quoted it.
template <int NACC>
__global__ __launch_bounds__(256) void kasm(double *out, int iters, double a, double b) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    const double av = a + 1e-3 * (threadIdx.x & 15), bv = b - 1e-3 * (threadIdx.x >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the other FP64 MFMA instruction: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction, 512 flop, one result per lane)
__global__ __launch_bounds__(256) void k4(double *out, int iters, double a, double b) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    const double av = a + 1e-3 * (threadIdx.x & 15), bv = b - 1e-3 * ((threadIdx.x & 63) >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same two instructions with operands that DIFFER from instruction to instruction (eight A and eight B registers of pseudo-random values):
// does the rate depend on the data / the operand registers?  which = 0: 16x16x4, 1: 4x4x4_4b
template <int WHICH>
__global__ __launch_bounds__(256) void kvar(double *out, int iters, const double *seed) {
    double a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed[(threadIdx.x * 17 + i * 5) & 1023]; b[i] = seed[(threadIdx.x * 29 + i * 11 + 3) & 1023]; }
    double s = 0;
    if constexpr (WHICH == 0) {
        v4d acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + it) & 7], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = 0.0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 7], b[(i >> 2)], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 4x4x4 MFMAs fed like the library's kernels feed them: per step twelve operands read from LDS (four A, eight B) for 32 MFMAs, the reads of step
// s + 1 issued before the MFMAs of step s (MODE 1), or read right before their use (MODE 0); MODE 2: no LDS at all, but the register copies of
// the pipelined loop.  What does an LDS-fed 4x4x4 loop sustain?
template <int MODE>
__global__ __launch_bounds__(256) void klds(double *out, int iters, const double *seed) {
    __shared__ double tile[64 * 34];
    for (int i = threadIdx.x; i < 64 * 34; i += 256) tile[i] = seed[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, arow = lane >> 4, acol = lane & 15;
    double acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    double an[4], bn[8];
    auto fetch = [&](int step) {
        const int ro = 4 * (step & 7) + arow;
#pragma unroll
        for (int m = 0; m < 4; ++m) an[m] = tile[(4 * m + (lane & 3)) * 34 + ro];
#pragma unroll
        for (int J = 0; J < 8; ++J) bn[J] = tile[((16 + 4 * J + acol) & 63) * 34 + ro];
    };
    fetch(0);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        double a[4], b[8];
        if constexpr (MODE == 0) fetch(it);
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = an[m];
#pragma unroll
        for (int J = 0; J < 8; ++J) b[J] = bn[J];
        if constexpr (MODE == 1) { fetch(it + 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int J = 0; J < 8; ++J)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[4 * J + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[m], b[J], acc[4 * J + m], 0, 0, 0);
        if constexpr (MODE == 1) __builtin_amdgcn_sched_barrier(0);
    }
    double s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i];
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
// MFMA and VALU side by side: the odd waves of every block issue v_fma_f64 chains, the even waves MFMAs -- does the chip deliver the SUM
// of the two sustained rates (separate pipes), or is there one budget (power / issue) that both draw from?
__global__ __launch_bounds__(256) void kmix(double *out, int iters_mfma, int iters_fma, double a, double b) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave & 1) {
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-9 + i;
        for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, 1e-9);
        }
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else {
        v4d acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
        const double av = a + 1e-3 * (threadIdx.x & 15), bv = b - 1e-3 * ((threadIdx.x & 63) >> 4);
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(double *d, unsigned long long *c, int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, c, iters, 1.0000001, 0.9999999);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long hc[8];
    hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    const double nm = (double)iters * NACC;                        // MFMAs per wave
    const double flop = (double)blocks * 4 * nm * 2048.0;
    // one block = 4 waves = one per SIMD; blocks_per_cu waves share a SIMD: cycles per MFMA and SIMD = block cycles / (nm * blocks_per_cu)
    const double cpm = (double)hc[0] / (nm * blocks_per_cu);
    const double mfma_per_simd = nm * blocks_per_cu;
    printf("{\"waves_per_simd\": %d, \"accumulators\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac_of_78.6\": %.3f, \"s_memtime_ticks_per_mfma_and_simd\": %.1f, "
           "\"mfma_per_simd_per_us\": %.1f}\n", blocks_per_cu, NACC, best, flop / best / 1e9, flop / best / 1e9 / 78.6, cpm, mfma_per_simd / (best * 1e3));
}
int main() {
    double *d; unsigned long long *c;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    hipMalloc(&c, sizeof(unsigned long long) * 256 * 8);
    const int iters = 4000;
    for (int bpc : {1, 2, 4, 8}) { run<4>(d, c, bpc, iters * 2); run<8>(d, c, bpc, iters); run<16>(d, c, bpc, iters / 2); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kfma, dim3(2048), dim3(256), 0, 0, d, 20000, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("{\"v_fma_f64\": true, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", ms, (double)2048 * 256 * 20000 * 16.0 * 2.0 / ms / 1e9);
    for (int bpc : {2, 8}) {
        const int it4 = 16000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k4, dim3(256 * bpc), dim3(256), 0, 0, d, it4, 1.0000001, 0.9999999);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("{\"v_mfma_f64_4x4x4_4b\": true, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", bpc, ms, (double)256 * bpc * 4 * it4 * 8.0 * 512.0 / ms / 1e9);
    }
    {
        double hs[1024];
        unsigned long long z = 88172645463325252ull;
        for (int i = 0; i < 1024; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; hs[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
        double *ds; hipMalloc(&ds, sizeof(hs)); hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
        for (int bpc : {2, 8}) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kvar<0>, dim3(256 * bpc), dim3(256), 0, 0, d, 4000, ds);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("{\"random_operands\": \"16x16x4\", \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", bpc, ms, (double)256 * bpc * 4 * 4000 * 8.0 * 2048.0 / ms / 1e9);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kvar<1>, dim3(256 * bpc), dim3(256), 0, 0, d, 4000, ds);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("{\"random_operands\": \"4x4x4_4b\", \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", bpc, ms, (double)256 * bpc * 4 * 4000 * 32.0 * 512.0 / ms / 1e9);
        }
    }
    {
        double hs2[1024];
        for (int i = 0; i < 1024; ++i) hs2[i] = 0.001 * (i % 97) - 0.04;
        double *ds2; hipMalloc(&ds2, sizeof(hs2)); hipMemcpy(ds2, hs2, sizeof(hs2), hipMemcpyHostToDevice);
        for (int bpc : {2, 4}) {
            for (int mode = 0; mode < 3; ++mode) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(klds<0>, dim3(256 * bpc), dim3(256), 0, 0, d, 4000, ds2);
                else if (mode == 1) hipLaunchKernelGGL(klds<1>, dim3(256 * bpc), dim3(256), 0, 0, d, 4000, ds2);
                else hipLaunchKernelGGL(klds<2>, dim3(256 * bpc), dim3(256), 0, 0, d, 4000, ds2);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                printf("{\"lds_fed_4x4x4\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n",
                       mode == 0 ? "12 reads right before 32 MFMAs" : (mode == 1 ? "reads of step s+1 before the MFMAs of step s" : "no LDS, register copies only"), bpc, ms,
                       (double)256 * bpc * 4 * 4000 * 32.0 * 512.0 / ms / 1e9);
            }
        }
    }
    for (int bpc : {1, 2, 4, 8}) {
        for (int nacc : {2, 4, 8}) {
            const int iters = 1000;
            hipEventRecord(e0);
            if (nacc == 2) hipLaunchKernelGGL(kasm<2>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0000001, 0.9999999);
            else if (nacc == 4) hipLaunchKernelGGL(kasm<4>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0000001, 0.9999999);
            else hipLaunchKernelGGL(kasm<8>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0000001, 0.9999999);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("{\"vgpr_form_inline_asm\": true, \"accumulators\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac_of_78.6\": %.3f}\n", nacc, bpc, ms,
                   (double)256 * bpc * 4 * iters * 64.0 * 2048.0 / ms / 1e9, (double)256 * bpc * 4 * iters * 64.0 * 2048.0 / ms / 1e9 / 78.6);
        }
    }
    for (int bpc : {1, 2, 4}) {
        for (int steps : {1, 8}) {
            const int iters = 8000 / steps;
            hipEventRecord(e0);
            if (steps == 1) hipLaunchKernelGGL(kpat<1>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0000001, 0.9999999);
            else hipLaunchKernelGGL(kpat<8>, dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0000001, 0.9999999);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("{\"product_shaped_16_accumulators\": true, \"k_steps_per_loop_iteration\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", steps, bpc, ms,
                   (double)256 * bpc * 4 * iters * steps * 16.0 * 2048.0 / ms / 1e9);
        }
    }
    // mixed: 8 blocks per CU, in every block two waves of MFMAs and two of FMAs, iteration counts chosen so that both halves take about as long alone
    for (int rep = 0; rep < 2; ++rep) {
        const int im = 4000, ifm = 13000;
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmix, dim3(2048), dim3(256), 0, 0, d, im, ifm, 1.0000001, 0.9999999);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double fm = (double)2048 * 2 * im * 8.0 * 2048.0, ff = (double)2048 * 128 * (double)ifm * 16.0 * 2.0;
        printf("{\"mixed_mfma_and_fma\": true, \"ms\": %.3f, \"MFMA_TFLOPs\": %.1f, \"FMA_TFLOPs\": %.1f, \"sum_TFLOPs\": %.1f}\n", ms, fm / ms / 1e9, ff / ms / 1e9,
               (fm + ff) / ms / 1e9);
    }
    return 0;
}
