// lk_kernels.hip.h -- gfx950 device kernels of the Krylov inner-loop engine.
//
// Two families on a column-contiguous panel (element (i, j) at X[j*ld + i]):
//   * the HBM-bandwidth bound BLAS-1/2 work of the graded path -- the abstract_vector primitives and the fused panel sweeps of the
//     (double) Gram-Schmidt step, ~1/6 flop/byte: VALU only, no MFMA.  Wavefront = 64 lanes; every global access is 16 bytes per lane
//     (1 KiB per wave instruction); reductions are register -> DPP wave sum -> LDS -> per-block partial -> fixed-order finish
//     kernel (deterministic, no float atomics);
//   * the tall-skinny CONTRACTIONS of the "next" rows (X Z of krylov_schur and the eigenvectors, X^H Y / X^H X with
//     many right-hand sides, the block Gram-Schmidt) on the FP64 matrix cores: __builtin_amdgcn_mfma_f64_16x16x4f64 tiles staged
//     through LDS (panel_gemm_mfma*, panel_xhy_mfma*, panel_gram_mfma*, panel_xhy_upd_mfma).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace lk {

typedef double v2d __attribute__((ext_vector_type(2)));

// ---- scalar traits -----------------------------------------------------------------
// Real kind: one v2d = two consecutive rows.  Complex kind: one v2d = (re, im) of one row.
template <bool CPLX> struct K;
template <> struct K<false> {
    static constexpr int ROWS = 2;  // rows per 16-byte lane access
    static constexpr int ELEM_DOUBLES = 1;
};
template <> struct K<true> {
    static constexpr int ROWS = 1;
    static constexpr int ELEM_DOUBLES = 2;
};

__device__ __forceinline__ v2d cmul(v2d a, v2d b) { return v2d{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ v2d cmulconj(v2d a, v2d b) {  // conj(a) * b
    return v2d{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x};
}

// Sum over the 64 lanes of a wave, result in EVERY lane.  Data-parallel-primitive moves inside the 16-lane rows
// (quad_perm x2, row_half_mirror, row_mirror: VALU cross-lane operands, no LDS crossbar) and four v_readlane for the rows:
// ~30 instructions, against ~1200 cycles for the six dependent ds_bpermute pairs of a __shfl_down tree -- which made the
// epilogue of a dot sweep cost 9 us (17 reductions per wave), a third of the whole kernel at launch-bound sizes.
// Fixed order: ((q0+q1)+(q2+q3)) within quads ... then (r0 + r1) + (r2 + r3) over the rows.  All lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);     // row_half_mirror
    v += dpp_move<0x140>(v);     // row_mirror: every lane holds its row's sum
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// The same DPP tree, but the 64 lanes are SC lane groups of 64/SC consecutive lanes (panel_sweep's lane split for wide
// bases): out[g] = sum over group g, wave-uniform.  SC = 1 is wave_sum.
template <int SC>
__device__ __forceinline__ void group_sums(double v, double (&out)[SC]) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    const double r0 = lane_value(v, 0), r1 = lane_value(v, 16), r2 = lane_value(v, 32), r3 = lane_value(v, 48);
    if constexpr (SC == 1) out[0] = (r0 + r1) + (r2 + r3);
    else if constexpr (SC == 2) { out[0] = r0 + r1; out[1] = r2 + r3; }
    else { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; }
}
// v + (the same value of the lanes 32 / 16 away): after it every lane group of a wave split SC ways holds the sum over the
// groups.  Addition is commutative, so all groups end with the same bits.
template <int SC>
__device__ __forceinline__ double across_groups(double v) {
    if constexpr (SC == 4) v += __shfl_xor(v, 16, 64);
    if constexpr (SC >= 2) v += __shfl_xor(v, 32, 64);
    return v;
}

// Early-exit guard of the ASYNCHRONOUS Arnoldi pipeline (lk_arnoldi enqueues every step without waiting for the
// host): when step s finds an invariant subspace / a colinear vector / a NaN, its normalise kernel records
// *stop_step = s; every kernel of a LATER step returns at once, so the basis beyond the breakdown stays untouched
// exactly as the reference leaves it (arnoldi.fypp:58-71).  stop_step == nullptr: no guard (synchronous callers).
struct Guard {
    const int *stop_step;
    int step;
};
__device__ __forceinline__ bool stopped(Guard g) {
    if (!g.stop_step) return false;
    const int s = *g.stop_step;          // uniform (scalar) load
    return s != 0 && s < g.step;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t seed, uint64_t ctr) {
    return (double)(splitmix64((seed << 32) + ctr) >> 11) * 0x1.0p-53;
}

// =====================================================================================
// Panel sweep: the kernel behind innerprod / linear_combination / orthogonalize / DGS.
//
//   UPDATE: y <- y - X(:, :k) * hin          (linear_combination + sub, gram_schmidt.fypp:144-145)
//   DOT   : partial h[j] += conj(X(:, j)) . y_out   (innerprod, gram_schmidt.fypp:141)
//   always: partial nrm2 += |y_out|^2        (the next pass's zero-vector check / qr's norm)
//
// A block of NW waves owns a tile of WR*64*ROWS rows x all k columns per iteration:
// lanes run along rows (coalesced 16 B/lane), the NW waves are split WC ways across the
// columns (each wave keeps <= KC columns of the tile in registers) and WR = NW/WC ways along
// rows.  With UPDATE the per-wave partial products meet in LDS (one barrier per tile, double
// buffered), and the SAME registers then feed the dot phase, so X is read from HBM exactly
// once per sweep.  Accumulators persist over the grid-stride tile loop; one shuffle+LDS
// reduction per block at the end.
//
// partial layout: partial[slot * pstride + blockIdx.x], slots 0..k-1 = h, slot k = nrm2.
// =====================================================================================
// One wave's slice of a tile: y (ROWS rows per lane) and the lane's rows of columns [0, nc) of Xw.
// `full` (block-uniform) selects the unguarded 16-byte path; the ragged last tile is guarded
// element-wise and zero filled.
template <bool CPLX>
__device__ __forceinline__ v2d load_y(const double *__restrict__ y, int64_t r, int64_t n, bool full) {
    if (full) return *reinterpret_cast<const v2d *>(y + r * K<CPLX>::ELEM_DOUBLES);
    if constexpr (CPLX) {
        return (r < n) ? *reinterpret_cast<const v2d *>(y + r * 2) : v2d{0.0, 0.0};
    } else {
        v2d yv;
        yv.x = (r < n) ? y[r] : 0.0;
        yv.y = (r + 1 < n) ? y[r + 1] : 0.0;
        return yv;
    }
}

// UNIFORM = false: nc differs between the lanes of a wave (lane-split sweeps): one predicated load per column, no
// wave-uniform fast path.
template <bool CPLX, int KC, bool UNIFORM = true>
__device__ __forceinline__ void load_cols(const double *__restrict__ Xw, int64_t colstride, int64_t r, int64_t n,
                                          bool full, int nc, v2d (&xv)[KC]) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    if (full) {
        if (UNIFORM && nc == KC) {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj)
                xv[jj] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(Xw + jj * colstride + r * ED));
        } else {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if (jj < nc)
                    xv[jj] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(Xw + jj * colstride + r * ED));
                else xv[jj] = v2d{0.0, 0.0};
            }
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            xv[jj] = v2d{0.0, 0.0};
            if (jj < nc) {
                const double *p = Xw + jj * colstride;
                if constexpr (CPLX) {
                    if (r < n) xv[jj] = *reinterpret_cast<const v2d *>(p + r * 2);
                } else {
                    if (r < n) xv[jj].x = p[r];
                    if (r + 1 < n) xv[jj].y = p[r + 1];
                }
            }
        }
    }
}

// Wide register tiles (KC > 16, SC = 1): the lane's rows of columns [0, nc) of a tile whose first row `Xt` points at (a
// wave-uniform pointer: scalar registers) -- each column's address is a scalar base plus ONE 32-bit lane offset
// (`global_load_dwordx4 v, v_off, s[base:base+1]`), where the generic path keeps a 64-bit VGPR address per column: 2 * KC
// vector registers that a 24- / 32-column tile does not have to spare.  `full` tiles only (the ragged tile takes load_cols).
template <int KC>
__device__ __forceinline__ void load_cols_sbase(const double *__restrict__ Xt, int64_t colstride, uint32_t lane_bytes, int nc,
                                                v2d (&xv)[KC]) {
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) {
        if (jj < nc) {
            const char *cb = reinterpret_cast<const char *>(Xt + jj * colstride);
            xv[jj] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(cb + lane_bytes));
        } else xv[jj] = v2d{0.0, 0.0};
    }
}

template <bool CPLX, int KC, bool UNIFORM = true>
__device__ __forceinline__ void load_tile(const double *__restrict__ Xw, int64_t colstride, const double *__restrict__ y,
                                          int64_t r, int64_t n, bool full, int nc, v2d (&xv)[KC], v2d &yv) {
    yv = load_y<CPLX>(y, r, n, full);
    load_cols<CPLX, KC, UNIFORM>(Xw, colstride, r, n, full, nc, xv);
}

// 16-byte store with an explicit gfx950 cache policy.  `policy` is block-uniform:
//   0 plain (write-back, line stays in the XCD's L2)   1 nt (streaming hint)
//   2 sc1 / 3 sc0 sc1 (write-through: the line leaves L2 in issue order instead of at eviction time)
__device__ __forceinline__ void store16(v2d *p, v2d v, int policy) {
    switch (policy) {
    case 1: __builtin_nontemporal_store(v, p); break;
    case 2: asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); break;
    case 3: asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); break;
    default: *p = v;
    }
}

template <bool CPLX>
__device__ __forceinline__ void store_rows(double *__restrict__ y, int64_t r, int64_t n, bool full, v2d yv,
                                           int policy = 0) {
    if (full) {
        store16(reinterpret_cast<v2d *>(y + r * K<CPLX>::ELEM_DOUBLES), yv, policy);
    } else if constexpr (CPLX) {
        if (r < n) *reinterpret_cast<v2d *>(y + r * 2) = yv;
    } else {
        if (r < n) y[r] = yv.x;
        if (r + 1 < n) y[r + 1] = yv.y;
    }
}

// TWO (sweep 3 of the two-pass DGS): two coefficient sets.  y' = y - X h1 is RE-formed from the
// original y exactly as sweep 2 formed it (same wave split, same summation order), then
// y'' = y' - X h2 is written.  That lets sweep 2 run with store = 0 (y' never goes to HBM): the
// y' write was ~1% of sweep 2's bytes but cost it 5-13% (HBM read/write mixing).
// SC (lane split, wide bases): the 64 lanes of a wave form SC groups of 64/SC lanes; group g of wave-column wc holds column
// group wc*SC + g for the SAME 64/SC * ROWS rows, so a block holds KC*NW*SC columns (256 / 512 for SC = 2 / 4) of a tile
// 1/SC as tall -- a column still arrives as contiguous 512 / 256 bytes per group.  That keeps the tile of X on chip between
// the update and the dot phase for up to 512 basis columns: ONE pass over X per sweep, 3k+4 columns per DGS, where column
// panels of 128 cost 4k - |last panel|.  The groups' partial products meet by two cross-lane adds before the LDS exchange,
// dots reduce per group.  SC = 1 is the narrow kernel, instruction for instruction.
// G (round 4; the two-coefficient sweep 3 of a lane-split DGS): a wave holds G COLUMN GROUPS of KC / G register slots each, all
// 64 lanes along rows (SC = 1) -- the column groups 'G wc .. G wc + G - 1' that the G lane groups of wave-column wc hold in the
// lane-split sweep 2 with the same (WC, kcw).  Every group's share of X h1 is summed over its own columns in slot order and the
// groups' shares are then added exactly as `across_groups` adds them, so y' = y - X h1 comes out BIT FOR BIT as sweep 2 formed it
// (the condition for dropping the y' store), while the tile is G times as tall and every lane has G times as many loads in
// flight: the update-only sweep has no accumulators to keep, so the registers the lane split spends on them hold columns instead.
template <bool CPLX, int KC, int NW, bool UPDATE, bool DOT, bool TWO, int SC = 1, int G = 1>
__global__ __launch_bounds__(NW * 64) void panel_sweep(const double *__restrict__ X, int64_t ldx, int k,
                                                        double *__restrict__ y, int64_t n,
                                                        const double *__restrict__ hin,
                                                        const double *__restrict__ hin2,
                                                        double *__restrict__ partial, int64_t pstride,
                                                        int WC, int kcw, int store, Guard guard) {
    static_assert(!TWO || (UPDATE && !DOT), "TWO is the update-only sweep with two coefficient sets");
    static_assert(SC == 1 || SC == 2 || SC == 4, "lane split");
    static_assert(G == 1 || G == 2 || G == 4, "column groups per wave");
    static_assert(G == 1 || (SC == 1 && UPDATE && !DOT && KC % G == 0), "column groups per wave: update-only sweeps without a lane split");
    if (stopped(guard)) return;
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int LG = 64 / SC;          // lanes per group
    constexpr int WROWS = LG * ROWS;     // rows one wave covers per tile
    constexpr int NU = TWO ? 2 : 1;
    constexpr bool BIG = KC > 16;        // wide register tile: scalar-base addressing (SC = 1), slim accumulators

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave % WC;
    const int wr = wave / WC;
    const int WR = NW / WC;
    const int lg = SC > 1 ? lane / LG : 0;            // lane group (per lane)
    const int rl = SC > 1 ? lane % LG : lane;         // lane within its group = row pair / row index
    constexpr int KCG = KC / G;          // register slots per column group
    const int c0 = (wc * SC + lg) * kcw * G;
    int nc = k - c0;
    nc = nc > kcw * G ? kcw * G : nc;
    nc = nc < 0 ? 0 : nc;

    __shared__ v2d u_lds[UPDATE ? 2 * NU * NW * 64 : 1];
    __shared__ double red_lds[NW * SC * (KC * ED + 1)];

    // projection coefficients of this wave's columns: wave-uniform (scalar registers) for SC = 1.  With the lane split they
    // differ from lane group to lane group and would take 2-4 VGPRs per column and coefficient set on top of the tile
    // itself (the complex two-coefficient sweep spilled): they live in LDS instead, zero padded to the block's capacity, and
    // are read back per use -- every lane of a group reads the same address (broadcast).
    constexpr bool HLDS = SC > 1 && UPDATE;
    constexpr int CAP = KC * NW * SC;
    __shared__ double hc_lds[HLDS ? NU * CAP * ED : 1];
    v2d hc[HLDS ? 1 : KC], hc2[(TWO && !HLDS) ? KC : 1];
    if constexpr (HLDS) {
        for (int i = threadIdx.x; i < CAP * ED; i += NW * 64) {
            hc_lds[i] = i < k * ED ? hin[i] : 0.0;
            if constexpr (TWO) hc_lds[CAP * ED + i] = i < k * ED ? hin2[i] : 0.0;
        }
        __syncthreads();
    } else if constexpr (UPDATE) {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            // slot jj holds column cj (G > 1: slot jj % KCG of column group jj / KCG, when that group has so many columns)
            const int cj = G > 1 ? c0 + (jj / KCG) * kcw + (jj % KCG) : c0 + jj;
            if (G > 1 ? ((jj % KCG) < kcw && cj < k) : (jj < nc)) {
                if constexpr (CPLX) hc[jj] = v2d{hin[2 * cj], hin[2 * cj + 1]};
                else hc[jj] = v2d{hin[cj], 0.0};
                if constexpr (TWO) {
                    if constexpr (CPLX) hc2[jj] = v2d{hin2[2 * cj], hin2[2 * cj + 1]};
                    else hc2[jj] = v2d{hin2[cj], 0.0};
                }
            } else {
                hc[jj] = v2d{0.0, 0.0};
                if constexpr (TWO) hc2[jj] = v2d{0.0, 0.0};
            }
        }
    }
    // coefficient jj of set `set` of this lane's column group (columns beyond k read the zero padding; c0 + jj < CAP always
    // holds for the groups that own columns, and the others are clamped onto the padding's last entry)
    auto hcoef = [&](int set, int jj) -> v2d {
        if constexpr (HLDS) {
            int idx = G > 1 ? c0 + (jj / KCG) * kcw + (jj % KCG) : c0 + jj;      // slot jj of group jj / KCG
            idx = idx < CAP ? idx : CAP - 1;
            // wide register tiles: keep the read INSIDE the tile loop (the index is laundered through an empty asm, so the
            // compiler cannot prove it loop invariant) -- hoisted, the NU * KC coefficients cost 2-4 VGPRs each on top of a tile
            // that already fills the register file (the real two-coefficient sweep at KC = 24 spilled 140 bytes per lane)
            if constexpr (BIG) asm volatile("" : "+v"(idx));
            if constexpr (CPLX) return *reinterpret_cast<const v2d *>(&hc_lds[(set * CAP + idx) * 2]);
            else return v2d{hc_lds[set * CAP + idx], 0.0};
        } else {
            if constexpr (TWO) return set ? hc2[jj] : hc[jj];
            else return hc[jj];
        }
    };

    // real: (sum over even rows, sum over odd rows) ; complex: (re, im).  The wide register tiles of the real kind (KC > 16)
    // keep ONE accumulator per column (both rows of the lane through the same chain): 2 * KC registers less.
    constexpr bool ACC1 = BIG && !CPLX;
    v2d acc[ACC1 ? 1 : KC];
    double acc1[ACC1 ? KC : 1];
#pragma unroll
    for (int jj = 0; jj < (ACC1 ? 1 : KC); ++jj) acc[jj] = v2d{0.0, 0.0};
#pragma unroll
    for (int jj = 0; jj < (ACC1 ? KC : 1); ++jj) acc1[jj] = 0.0;
    double nrm = 0.0;

    const int64_t tile_rows = (int64_t)WR * WROWS;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const double *Xw = X + (int64_t)c0 * ldx * ED;
    const int64_t colstride = ldx * ED;  // doubles between consecutive columns
    const int64_t roff = (int64_t)wr * WROWS + (int64_t)rl * ROWS;
    int buf = 0;

    // tile order.  Default: cyclic over the whole grid (block b takes tiles b, b + G, ...).  Bit 4 of `store` (A/B knob
    // "xcd_map"): workgroups are dealt round-robin to the 8 XCDs, so give XCD x (= blockIdx % 8) a CONTIGUOUS eighth of the
    // rows and let its blocks walk it cyclically -- each XCD's L2 then streams one address range per column.
    const bool xmap = (store & 16) && gridDim.x >= 8 && (gridDim.x & 7) == 0;
    const int64_t chunk = xmap ? (ntiles + 7) / 8 : ntiles;
    const int64_t tbase = xmap ? (int64_t)(blockIdx.x & 7) * chunk : 0;
    const int64_t tend = xmap ? (tbase + chunk < ntiles ? tbase + chunk : ntiles) : ntiles;
    const int64_t tstep = xmap ? gridDim.x >> 3 : gridDim.x;
    for (int64_t t = tbase + (xmap ? blockIdx.x >> 3 : blockIdx.x); t < tend; t += tstep) {
        const int64_t r = t * tile_rows + roff;        // first row of this lane
        const bool full = (t + 1) * tile_rows <= n;   // block-uniform
        v2d xv[KC];
        v2d yv;
        if constexpr (G > 1) {
            // slot g * KCG + j <- column c0 + g * kcw + j (j < kcw, column < k); the other slots hold zeros
            yv = load_y<CPLX>(y, r, n, full);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                int ng = nc - g * kcw;
                ng = ng > kcw ? kcw : ng;
                if (full) {
                    v2d xg[KCG];
                    load_cols_sbase<KCG>(Xw + (int64_t)g * kcw * colstride + (t * tile_rows + (int64_t)wr * WROWS) * ED, colstride,
                                         (uint32_t)(rl * ROWS * ED * 8), ng, xg);
#pragma unroll
                    for (int j = 0; j < KCG; ++j) xv[g * KCG + j] = xg[j];
                } else {
                    v2d xg[KCG];
                    load_cols<CPLX, KCG, false>(Xw + (int64_t)g * kcw * colstride, colstride, r, n, false, ng < 0 ? 0 : ng, xg);
#pragma unroll
                    for (int j = 0; j < KCG; ++j) xv[g * KCG + j] = xg[j];
                }
            }
        } else if constexpr (BIG && SC == 1) {
            if (full) {
                yv = load_y<CPLX>(y, r, n, true);
                load_cols_sbase<KC>(Xw + (t * tile_rows + (int64_t)wr * WROWS) * ED, colstride, (uint32_t)(rl * ROWS * ED * 8), nc, xv);
            } else {
                load_tile<CPLX, KC, false>(Xw, colstride, y, r, n, false, nc, xv, yv);
            }
        } else {
            load_tile<CPLX, KC, SC == 1>(Xw, colstride, y, r, n, full, nc, xv, yv);
        }

        if constexpr (UPDATE) {
            v2d u = v2d{0.0, 0.0}, u2 = v2d{0.0, 0.0};
            if constexpr (G > 1) {
                // per column group in slot order, then the groups' shares in the order across_groups adds them:
                // (g0 + g1) for two groups, ((g0 + g1) + (g2 + g3)) for four
                v2d ug[G], ug2[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    ug[g] = v2d{0.0, 0.0};
                    ug2[g] = v2d{0.0, 0.0};
#pragma unroll
                    for (int j = 0; j < KCG; ++j) {
                        const int jj = g * KCG + j;
                        if constexpr (CPLX) ug[g] += cmul(xv[jj], hcoef(0, jj));
                        else ug[g] += xv[jj] * hcoef(0, jj).x;
                    }
                    if constexpr (TWO) {
#pragma unroll
                        for (int j = 0; j < KCG; ++j) {
                            const int jj = g * KCG + j;
                            if constexpr (CPLX) ug2[g] += cmul(xv[jj], hcoef(1, jj));
                            else ug2[g] += xv[jj] * hcoef(1, jj).x;
                        }
                    }
                }
                if constexpr (G == 2) { u = ug[0] + ug[1]; u2 = ug2[0] + ug2[1]; }
                else { u = (ug[0] + ug[1]) + (ug[2] + ug[3]); u2 = (ug2[0] + ug2[1]) + (ug2[2] + ug2[3]); }
            } else {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if constexpr (CPLX) u += cmul(xv[jj], hcoef(0, jj));
                else u += xv[jj] * hcoef(0, jj).x;
            }
            if constexpr (TWO) {
#pragma unroll
                for (int jj = 0; jj < KC; ++jj) {
                    if constexpr (CPLX) u2 += cmul(xv[jj], hcoef(1, jj));
                    else u2 += xv[jj] * hcoef(1, jj).x;
                }
            }
            }
            if constexpr (SC > 1) {                    // the lane groups of this wave hold the same rows: add their shares
                u.x = across_groups<SC>(u.x);
                u.y = across_groups<SC>(u.y);
                if constexpr (TWO) {
                    u2.x = across_groups<SC>(u2.x);
                    u2.y = across_groups<SC>(u2.y);
                }
            }
            if (WC > 1) {
                v2d *ub = u_lds + buf * (NU * NW * 64);
                ub[wave * 64 + lane] = u;
                if constexpr (TWO) ub[NW * 64 + wave * 64 + lane] = u2;
                __syncthreads();
                v2d s = v2d{0.0, 0.0}, s2 = v2d{0.0, 0.0};
                for (int w = 0; w < WC; ++w) s += ub[(wr * WC + w) * 64 + lane];
                if constexpr (TWO)
                    for (int w = 0; w < WC; ++w) s2 += ub[NW * 64 + (wr * WC + w) * 64 + lane];
                u = s;
                u2 = s2;
                buf ^= 1;
            }
            yv -= u;
            if constexpr (TWO) yv -= u2;
            // `store`: 0 = keep y' in registers; otherwise bit 0 set, bits 1-2 = cache policy of the 16-B
            // store, bit 3 = every wave of the column split stores its own 64/WC-lane slice of the rows
            // (instead of the wc == 0 wave storing all 64 lanes).
            if (store & 1) {
                const int pol = (store >> 1) & 3;
                if constexpr (SC > 1) {
                    if (wc == 0 && lg == 0) store_rows<CPLX>(y, r, n, full, yv, pol);
                } else if (store & 8) {
                    const int per = 64 / WC;
                    if (lane / per == wc) store_rows<CPLX>(y, r, n, full, yv, pol);
                } else if (wc == 0) {
                    store_rows<CPLX>(y, r, n, full, yv, pol);
                }
            }
        }
        if constexpr (DOT) {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if constexpr (CPLX) acc[jj] += cmulconj(xv[jj], yv);
                else if constexpr (ACC1) acc1[jj] = fma(xv[jj].y, yv.y, fma(xv[jj].x, yv.x, acc1[jj]));
                else acc[jj] += xv[jj] * yv;
            }
        }
        if (wc == 0 && lg == 0) nrm += yv.x * yv.x + yv.y * yv.y;
    }

    // ---- block reduction: lanes (shuffle) -> waves sharing a column set (LDS) -> partial
    constexpr int SLOTS = KC * ED + 1;
    if constexpr (DOT) {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            if constexpr (SC == 1) {
                if constexpr (CPLX) {
                    double re = wave_sum(acc[jj].x), im = wave_sum(acc[jj].y);
                    if (lane == 0) { red_lds[wave * SLOTS + 2 * jj] = re; red_lds[wave * SLOTS + 2 * jj + 1] = im; }
                } else {
                    double s;
                    if constexpr (ACC1) s = wave_sum(acc1[jj]);
                    else s = wave_sum(acc[jj].x + acc[jj].y);
                    if (lane == 0) red_lds[wave * SLOTS + jj] = s;
                }
            } else {
                double a[SC], b[SC];
                if constexpr (CPLX) {
                    group_sums<SC>(acc[jj].x, a);
                    group_sums<SC>(acc[jj].y, b);
                } else if constexpr (ACC1) {
                    group_sums<SC>(acc1[jj], a);
                } else {
                    group_sums<SC>(acc[jj].x + acc[jj].y, a);
                }
                if (lane == 0) {
#pragma unroll
                    for (int gq = 0; gq < SC; ++gq) {
                        if constexpr (CPLX) {
                            red_lds[(wave * SC + gq) * SLOTS + 2 * jj] = a[gq];
                            red_lds[(wave * SC + gq) * SLOTS + 2 * jj + 1] = b[gq];
                        } else {
                            red_lds[(wave * SC + gq) * SLOTS + jj] = a[gq];
                        }
                    }
                }
            }
        }
    }
    {
        double s = wave_sum(nrm);
        if (lane == 0) red_lds[(wave * SC) * SLOTS + KC * ED] = s;
    }
    __syncthreads();
    // thread tid < k*ED handles one output double: column j = tid/ED lives in column group j / kcw
    const int tid = threadIdx.x;
    if constexpr (DOT) {
        for (int o = tid; o < k * ED; o += NW * 64) {
            const int j = o / ED, part = o % ED;
            const int cgj = j / kcw, jj = j - cgj * kcw;
            double s = 0.0;
            for (int w = 0; w < WR; ++w) s += red_lds[(w * WC * SC + cgj) * SLOTS + jj * ED + part];
            partial[((int64_t)j * ED + part) * pstride + blockIdx.x] = s;
        }
    }
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < WR; ++w) s += red_lds[(w * WC * SC) * SLOTS + KC * ED];
        partial[((int64_t)k * ED) * pstride + blockIdx.x] = s;
        if constexpr (CPLX) partial[((int64_t)k * ED + 1) * pstride + blockIdx.x] = 0.0;
    }
}

// Dot sweep, ONE COLUMN AT A TIME (DGS sweep 1, innerprod): h[j] = conj(X(:, j)) . y and ||y||^2.
// X is used once and nothing has to meet across columns, so a block can keep its rows of y in registers and walk the k
// columns one after the other, reading U wave-instructions = U KiB of contiguous rows per wave from each: long contiguous
// runs per column instead of 1 KiB per column per wave.  tools/colwise_probe.hip: 7.0 TB/s against 6.8 for every shape of
// the all-columns-per-tile sweep (tools/sweep_probe.hip).  Per column and tile a wave reduces its partial sum with the DPP
// wave_sum and its lane 0 adds it to the wave's own LDS slot: fixed order, no atomics; the 4 waves' slots meet at the end.
// partial layout as panel_sweep: slot j (ED doubles) = h[j], slot k = the norm.
template <bool CPLX, int U>
__global__ __launch_bounds__(256) void panel_dot_cw(const double *__restrict__ X, int64_t ldx, int k,
                                                    const double *__restrict__ y, int64_t n, double *__restrict__ partial,
                                                    int64_t pstride, Guard guard) {
    if (stopped(guard)) return;
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int SEG = 256 * ROWS;                 // rows one block-wide 16-byte access covers
    extern __shared__ double acc_lds[];             // [4 waves][(k + 1) * ED]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nslot = (k + 1) * ED;
    double *mine = acc_lds + wave * nslot;
    for (int i = lane; i < nslot; i += 64) mine[i] = 0.0;
    const int64_t tile_rows = (int64_t)SEG * U;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const int64_t colstride = ldx * ED;
    double nrm = 0.0;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r0 = t * tile_rows + (int64_t)threadIdx.x * ROWS;
        const bool full = (t + 1) * tile_rows <= n;
        v2d yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            yv[u] = load_y<CPLX>(y, r0 + (int64_t)u * SEG, n, full);
            nrm += yv[u].x * yv[u].x + yv[u].y * yv[u].y;
        }
        for (int j = 0; j < k; ++j) {
            const double *xc = X + (int64_t)j * colstride;
            v2d xv[U];
            if (full) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    xv[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + (r0 + (int64_t)u * SEG) * ED));
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = load_y<CPLX>(xc, r0 + (int64_t)u * SEG, n, false);
            }
            if constexpr (CPLX) {
                double re = 0.0, im = 0.0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const v2d z = cmulconj(xv[u], yv[u]);
                    re += z.x;
                    im += z.y;
                }
                re = wave_sum(re);
                im = wave_sum(im);
                if (lane == 0) { mine[2 * j] += re; mine[2 * j + 1] += im; }
            } else {
                double sm = 0.0;
#pragma unroll
                for (int u = 0; u < U; ++u) sm = fma(xv[u].y, yv[u].y, fma(xv[u].x, yv[u].x, sm));
                sm = wave_sum(sm);
                if (lane == 0) mine[j] += sm;
            }
        }
    }
    {
        const double sn = wave_sum(nrm);
        if (lane == 0) { mine[k * ED] = sn; if constexpr (CPLX) mine[k * ED + 1] = 0.0; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nslot; i += 256)
        partial[(int64_t)i * pstride + blockIdx.x] = (acc_lds[i] + acc_lds[nslot + i]) + (acc_lds[2 * nslot + i] + acc_lds[3 * nslot + i]);
}

// Multi-right-hand-side dots: M(:, q) = X(:, :k)^H Y(:, q) and ||Y(:, q)||^2 for P columns of Y in ONE pass
// over X (innerprod_matrix, AbstractVectors.fypp:677-695; Gram; the dot sweeps of DGS_basis_against_basis,
// gram_schmidt.fypp:59-105).  Same wave split as panel_sweep<DOT>; P accumulator sets per column.
// partial layout: slot (q*(k+1) + j)*ED (+part), j = k is the norm slot of column q.
template <bool CPLX, int KC, int NW, int P>
__global__ __launch_bounds__(NW * 64) void panel_dot_p(const double *__restrict__ X, int64_t ldx, int k,
                                                        const double *__restrict__ Y, int64_t ldy, int pn, int64_t n,
                                                        double *__restrict__ partial, int64_t pstride, int WC, int kcw,
                                                        int kslots, int j0, int with_norm) {
    // This launch covers columns [j0, j0 + k) of a basis of kslots - 1 columns: results go to slot
    // (q * kslots + j0 + j) * ED (+part); the norm of Y(:, q) to slot (q * kslots + kslots - 1) * ED when with_norm.
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int WROWS = 64 * ROWS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave % WC, wr = wave / WC, WR = NW / WC;
    const int c0 = wc * kcw;
    int nc = k - c0;
    nc = nc > kcw ? kcw : nc;
    nc = nc < 0 ? 0 : nc;
    constexpr int SLOTS = (KC * ED + 1) * P;
    // real kind, four right-hand sides: ONE accumulator per (column, right-hand side) fed by a two-FMA chain over the lane's two rows (as
    // panel_sweep_p does) -- sixteen two-row accumulators beside the tile do not fit the 128 registers of a 1024-thread block (12 B of scratch)
    constexpr bool CHAIN = !CPLX && P == 4;
    __shared__ double red_lds[NW * SLOTS];

    v2d acc[P][KC];
    double nrm[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        nrm[q] = 0.0;
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) acc[q][jj] = v2d{0.0, 0.0};
    }
    const int64_t tile_rows = (int64_t)WR * WROWS;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const double *Xw = X + (int64_t)c0 * ldx * ED;
    const int64_t colstride = ldx * ED, ystride = ldy * ED;
    const int64_t roff = (int64_t)wr * WROWS + (int64_t)lane * ROWS;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r = t * tile_rows + roff;
        const bool full = (t + 1) * tile_rows <= n;
        v2d xv[KC], yv[P];
        load_cols<CPLX, KC>(Xw, colstride, r, n, full, nc, xv);
#pragma unroll
        for (int q = 0; q < P; ++q) yv[q] = (q < pn) ? load_y<CPLX>(Y + q * ystride, r, n, full) : v2d{0.0, 0.0};
#pragma unroll
        for (int q = 0; q < P; ++q) {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if constexpr (CPLX) acc[q][jj] += cmulconj(xv[jj], yv[q]);
                else if constexpr (CHAIN) acc[q][jj].x = fma(xv[jj].y, yv[q].y, fma(xv[jj].x, yv[q].x, acc[q][jj].x));
                else acc[q][jj] += xv[jj] * yv[q];
            }
            if (wc == 0) nrm[q] += yv[q].x * yv[q].x + yv[q].y * yv[q].y;
        }
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
        double *rl = red_lds + wave * SLOTS + q * (KC * ED + 1);
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            if constexpr (CPLX) {
                double re = wave_sum(acc[q][jj].x), im = wave_sum(acc[q][jj].y);
                if (lane == 0) { rl[2 * jj] = re; rl[2 * jj + 1] = im; }
            } else {
                double sm = wave_sum(CHAIN ? acc[q][jj].x : acc[q][jj].x + acc[q][jj].y);
                if (lane == 0) rl[jj] = sm;
            }
        }
        double sn = wave_sum(nrm[q]);
        if (lane == 0) rl[KC * ED] = sn;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    for (int idx = tid; idx < P * (k + 1) * ED; idx += blockDim.x) {
        const int q = idx / ((k + 1) * ED), rem = idx % ((k + 1) * ED);
        const int j = rem / ED, part = rem % ED;
        double sm = 0.0;
        int64_t slot;
        if (j < k) {
            const int wcj = j / kcw, jj = j - wcj * kcw;
            for (int w = 0; w < WR; ++w) sm += red_lds[(w * WC + wcj) * SLOTS + q * (KC * ED + 1) + jj * ED + part];
            slot = ((int64_t)q * kslots + j0 + j) * ED + part;
        } else {
            if (!with_norm) continue;
            if (part == 0)
                for (int w = 0; w < WR; ++w) sm += red_lds[(w * WC) * SLOTS + q * (KC * ED + 1) + KC * ED];
            slot = ((int64_t)q * kslots + kslots - 1) * ED + part;
        }
        partial[slot * pstride + blockIdx.x] = sm;
    }
}

// The fused sweeps of the DGS for up to P columns of Y at once (DGS_basis_against_basis, gram_schmidt.fypp:59-105):
//   UPDATE + DOT       : Y' = Y - X H1 (kept in registers unless `store`), H2 = X^H Y', ||Y'_q||^2      (pass B)
//   UPDATE + TWO       : Y'' = (Y - X H1) - X H2, stored                                                 (pass C)
// Same wave split and summation order in both, so pass C re-forms exactly the Y' that pass B projected (see panel_sweep).
// One pass over X serves all P columns: with the multi-right-hand-side dot pass in front (panel_dot_p) a block DGS
// costs THREE passes over X per group of P columns, against four for dots / update / dots / update.
// Coefficients and results use panel_dot_p's layout: slot (q * kslots + j) * ED (+part), j = kslots - 1 = the norm slot.
// One accumulator per (column, right-hand side) -- real: a two-FMA chain over the lane's two rows -- and a single LDS
// exchange buffer with two barriers per tile keep P = 4 x 16 columns inside the register file and 64 KB of LDS.
template <bool CPLX, int KC, int NW, int P, bool DOT, bool TWO>
__global__ __launch_bounds__(NW * 64) void panel_sweep_p(const double *__restrict__ X, int64_t ldx, int k,
                                                          double *__restrict__ Y, int64_t ldy, int pn, int64_t n,
                                                          const double *__restrict__ hin, const double *__restrict__ hin2,
                                                          int kslots, double *__restrict__ partial, int64_t pstride, int WC,
                                                          int kcw, int store, Guard guard) {
    static_assert(!(DOT && TWO), "pass B carries the dots, pass C the second coefficient set");
    if (stopped(guard)) return;
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int WROWS = 64 * ROWS;
    constexpr int NU = TWO ? 2 : 1;
    constexpr int AD = CPLX ? 2 : 1;                    // doubles per accumulator
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave % WC, wr = wave / WC, WR = NW / WC;
    const int c0 = wc * kcw;
    int nc = k - c0;
    nc = nc > kcw ? kcw : nc;
    nc = nc < 0 ? 0 : nc;
    __shared__ v2d u_lds[NU * P * NW * 64];
    constexpr int SLOTS = (KC * ED + 1) * P;
    __shared__ double red_lds[DOT ? NW * SLOTS : 1];

    double acc[P][KC][AD];
    double nrm[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        nrm[q] = 0.0;
#pragma unroll
        for (int jj = 0; jj < KC; ++jj)
#pragma unroll
            for (int a = 0; a < AD; ++a) acc[q][jj][a] = 0.0;
    }
    const int64_t tile_rows = (int64_t)WR * WROWS;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const double *Xw = X + (int64_t)c0 * ldx * ED;
    const int64_t colstride = ldx * ED, ystride = ldy * ED;
    const int64_t roff = (int64_t)wr * WROWS + (int64_t)lane * ROWS;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r = t * tile_rows + roff;
        const bool full = (t + 1) * tile_rows <= n;
        v2d xv[KC], yv[P];
        load_cols<CPLX, KC>(Xw, colstride, r, n, full, nc, xv);
#pragma unroll
        for (int q = 0; q < P; ++q) yv[q] = (q < pn) ? load_y<CPLX>(Y + q * ystride, r, n, full) : v2d{0.0, 0.0};
        // this wave's share of X H for every right-hand side (coefficients are wave-uniform: scalar loads)
        v2d u[NU][P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            u[0][q] = v2d{0.0, 0.0};
            if constexpr (TWO) u[1][q] = v2d{0.0, 0.0};
            const double *h1 = hin + ((int64_t)q * kslots + c0) * ED;
            const double *h2 = TWO ? hin2 + ((int64_t)q * kslots + c0) * ED : nullptr;
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if (jj < nc && q < pn) {
                    if constexpr (CPLX) {
                        u[0][q] += cmul(xv[jj], v2d{h1[2 * jj], h1[2 * jj + 1]});
                        if constexpr (TWO) u[1][q] += cmul(xv[jj], v2d{h2[2 * jj], h2[2 * jj + 1]});
                    } else {
                        u[0][q] += xv[jj] * h1[jj];
                        if constexpr (TWO) u[1][q] += xv[jj] * h2[jj];
                    }
                }
            }
        }
        if (WC > 1) {
#pragma unroll
            for (int s = 0; s < NU; ++s)
#pragma unroll
                for (int q = 0; q < P; ++q) u_lds[((s * P + q) * NW + wave) * 64 + lane] = u[s][q];
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NU; ++s)
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    v2d sum = v2d{0.0, 0.0};
                    for (int w = 0; w < WC; ++w) sum += u_lds[((s * P + q) * NW + wr * WC + w) * 64 + lane];
                    u[s][q] = sum;
                }
            __syncthreads();                                    // single buffer: everyone has read before the next tile writes
        }
#pragma unroll
        for (int q = 0; q < P; ++q) {
            yv[q] -= u[0][q];
            if constexpr (TWO) yv[q] -= u[1][q];
            if (store && wc == 0 && q < pn) store_rows<CPLX>(Y + q * ystride, r, n, full, yv[q], (store >> 1) & 3);
            if constexpr (DOT) {
#pragma unroll
                for (int jj = 0; jj < KC; ++jj) {
                    if constexpr (CPLX) {
                        const v2d z = cmulconj(xv[jj], yv[q]);
                        acc[q][jj][0] += z.x;
                        acc[q][jj][1] += z.y;
                    } else {
                        acc[q][jj][0] = fma(xv[jj].y, yv[q].y, fma(xv[jj].x, yv[q].x, acc[q][jj][0]));
                    }
                }
                if (wc == 0) nrm[q] += yv[q].x * yv[q].x + yv[q].y * yv[q].y;
            }
        }
    }
    if constexpr (DOT) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            double *rl = red_lds + wave * SLOTS + q * (KC * ED + 1);
#pragma unroll
            for (int jj = 0; jj < KC; ++jj)
#pragma unroll
                for (int a = 0; a < AD; ++a) {
                    const double sm = wave_sum(acc[q][jj][a]);
                    if (lane == 0) rl[jj * ED + a] = sm;
                }
            const double sn = wave_sum(nrm[q]);
            if (lane == 0) rl[KC * ED] = sn;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < P * (k + 1) * ED; idx += blockDim.x) {
            const int q = idx / ((k + 1) * ED), rem = idx % ((k + 1) * ED);
            const int j = rem / ED, part = rem % ED;
            double sm = 0.0;
            int64_t slot;
            if (j < k) {
                const int wcj = j / kcw, jj = j - wcj * kcw;
                for (int w = 0; w < WR; ++w) sm += red_lds[(w * WC + wcj) * SLOTS + q * (KC * ED + 1) + jj * ED + part];
                slot = ((int64_t)q * kslots + j) * ED + part;
            } else {
                if (part == 0)
                    for (int w = 0; w < WR; ++w) sm += red_lds[(w * WC) * SLOTS + q * (KC * ED + 1) + KC * ED];
                slot = ((int64_t)q * kslots + kslots - 1) * ED + part;
            }
            partial[slot * pstride + blockIdx.x] = sm;
        }
    }
}

// Streaming update y <- y - X(:, :k) * hin with ||y_out||^2: no dots, so nothing has to stay in
// registers and every wave can walk ALL k columns of its own rows in chunks of KC -- no LDS
// exchange, no barrier in the loop.  Used for DGS sweep 3 and for linear_combination.
template <bool CPLX, int KC, int NW, bool TWO>
__global__ __launch_bounds__(NW * 64) void panel_update(const double *__restrict__ X, int64_t ldx, int k,
                                                         double *__restrict__ y, int64_t n,
                                                         const double *__restrict__ hin,
                                                         const double *__restrict__ hin2,
                                                         double *__restrict__ partial, int64_t pstride, int policy,
                                                         Guard guard) {
    if (stopped(guard)) return;
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int WROWS = 64 * ROWS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tile_rows = (int64_t)NW * WROWS;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const int64_t colstride = ldx * ED;
    double nrm = 0.0;
    __shared__ double red_lds[NW];

    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r = t * tile_rows + (int64_t)wave * WROWS + (int64_t)lane * ROWS;
        const bool full = (t + 1) * tile_rows <= n;
        v2d u = v2d{0.0, 0.0}, u2 = v2d{0.0, 0.0};
        v2d yv = load_y<CPLX>(y, r, n, full);
        for (int c0 = 0; c0 < k; c0 += KC) {
            int nc = k - c0;
            nc = nc > KC ? KC : nc;
            v2d xv[KC];
            load_cols<CPLX, KC>(X + (int64_t)c0 * colstride, colstride, r, n, full, nc, xv);
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if (jj < nc) {
                    if constexpr (CPLX) u += cmul(xv[jj], v2d{hin[2 * (c0 + jj)], hin[2 * (c0 + jj) + 1]});
                    else u += xv[jj] * hin[c0 + jj];
                    if constexpr (TWO) {
                        if constexpr (CPLX) u2 += cmul(xv[jj], v2d{hin2[2 * (c0 + jj)], hin2[2 * (c0 + jj) + 1]});
                        else u2 += xv[jj] * hin2[c0 + jj];
                    }
                }
            }
        }
        yv -= u;
        if constexpr (TWO) yv -= u2;      // y'' = (y - X h1) - X h2
        store_rows<CPLX>(y, r, n, full, yv, policy);
        nrm += yv.x * yv.x + yv.y * yv.y;
    }
    const double s = wave_sum(nrm);
    if (lane == 0) red_lds[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < NW; ++w) tot += red_lds[w];
        partial[((int64_t)k * ED) * pstride + blockIdx.x] = tot;
        if constexpr (CPLX) partial[((int64_t)k * ED + 1) * pstride + blockIdx.x] = 0.0;
    }
}

// Tall-skinny product Y(:, :qn) (+)= X(:, :k) * C  -- linear_combination_matrix (AbstractVectors.fypp:605-643)
// in ONE pass over X for up to NQG*QB output columns, instead of k*q axpbys.  Used by krylov_schur's basis update
// X <- X Z (BaseKrylov.fypp:816-824), eigs' eigenvector reconstruction (IterativeSolvers.fypp:1127-1132), the
// GMRES solution update (gmres.fypp:201) and the block Gram-Schmidt update.
//
// FP64-FMA bound for the complex kind (8 k q flop per 16 (k+q) bytes), balanced for the real kind, so the design
// goal is back-to-back v_fma_f64 issue: lanes run along rows (16 B per lane), every lane keeps QB accumulators in
// VGPRs, and the coefficients are WAVE-UNIFORM: they come in through scalar loads (SGPRs feed the FMA directly --
// no LDS, no VGPR broadcast).  The 4 waves of a block split the output columns QGB ways (QGB = 1, 2, 4 groups of
// QB) and the rows 4/QGB ways; waves of one block that share rows read the same X lines at the same time, so
// X leaves HBM once (the repeats are L1/L2 hits).
// Cp: coefficients packed by the host as [group g][column j < k][qq < QB][ED], zero padded, sign folded in.
// acc[qq] += X(r, j0 + jj) * C(j0 + jj, qq) for NCOL consecutive columns; GUARD = element-wise guarded loads
// (ragged last tile).  cj points at the packed coefficients of column j0 (uniform -> scalar loads).
template <bool CPLX, int NCOL, int QB, bool GUARD>
__device__ __forceinline__ void gemm_cols(const double *__restrict__ Xj, int64_t xstride, int64_t r, int64_t n,
                                          const double *__restrict__ cj, v2d (&acc)[QB]) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    v2d xv[NCOL];
    load_cols<CPLX, NCOL>(Xj, xstride, r, n, !GUARD, NCOL, xv);
#pragma unroll
    for (int jj = 0; jj < NCOL; ++jj) {
        const double *__restrict__ c = cj + jj * (QB * ED);
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) {
            if constexpr (CPLX) {                                   // 4 FMAs, coefficient operands in SGPRs
                const double cr = c[2 * qq], ci = c[2 * qq + 1];
                acc[qq].x = fma(xv[jj].x, cr, acc[qq].x);
                acc[qq].x = fma(-xv[jj].y, ci, acc[qq].x);
                acc[qq].y = fma(xv[jj].x, ci, acc[qq].y);
                acc[qq].y = fma(xv[jj].y, cr, acc[qq].y);
            } else {
                const double cr = c[qq];
                acc[qq].x = fma(xv[jj].x, cr, acc[qq].x);
                acc[qq].y = fma(xv[jj].y, cr, acc[qq].y);
            }
        }
    }
}

template <bool CPLX, int KC, int QB>
__global__ __launch_bounds__(256) void panel_gemm(const double *__restrict__ X, int64_t ldx, int k,
                                                  double *__restrict__ Y, int64_t ldy, int qn,
                                                  const double *__restrict__ Cp, int64_t n, int accumulate, int QGB,
                                                  int policy, Guard guard) {
    if (stopped(guard)) return;
    // QB = accumulators per lane = output columns per wave: 16 for wide products; 1 / 2 / 4 / 8 for narrow ones, where the
    // kernel is a pure stream over X (q = 1: the GMRES solution update, gmres.fypp:201, and every X * v) -- with QB = 16 a
    // single output column paid 16x the FMAs and ran VALU-limited at 4.9 TB/s.
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int WROWS = 64 * ROWS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qg = wave % QGB, rs = wave / QGB, RS = 4 / QGB;
    const int q0 = qg * QB;
    int nq = qn - q0;
    nq = nq > QB ? QB : nq;
    if (nq <= 0) return;                                     // no barrier in this kernel
    const double *__restrict__ Cw = Cp + (int64_t)qg * k * (QB * ED);
    const int tile_rows = RS * WROWS;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const int64_t xstride = ldx * ED, ystride = ldy * ED;
    const int kfast = (k / KC) * KC;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t r = t * tile_rows + (int64_t)rs * WROWS + (int64_t)lane * ROWS;
        const bool full = (t + 1) * tile_rows <= n;
        v2d acc[QB];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) acc[qq] = v2d{0.0, 0.0};
        if (full) {
            for (int c0 = 0; c0 < kfast; c0 += KC)
                gemm_cols<CPLX, KC, QB, false>(X + (int64_t)c0 * xstride, xstride, r, n, Cw + (int64_t)c0 * (QB * ED), acc);
            for (int c0 = kfast; c0 < k; ++c0)
                gemm_cols<CPLX, 1, QB, false>(X + (int64_t)c0 * xstride, xstride, r, n, Cw + (int64_t)c0 * (QB * ED), acc);
        } else {
            for (int c0 = 0; c0 < k; ++c0)
                gemm_cols<CPLX, 1, QB, true>(X + (int64_t)c0 * xstride, xstride, r, n, Cw + (int64_t)c0 * (QB * ED), acc);
        }
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) {
            if (qq < nq) {
                double *yc = Y + (int64_t)(q0 + qq) * ystride;
                v2d out = acc[qq];
                if (accumulate) out += load_y<CPLX>(yc, r, n, full);
                store_rows<CPLX>(yc, r, n, full, out, policy);
            }
        }
    }
}

// ---- FP64-MFMA tall-skinny product --------------------------------------------------------------------------------
// Same contract as panel_gemm for k <= 128, on the matrix cores: v_mfma_f64_16x16x4_f64 computes D(16x16) += A(16x4) B(4x16)
// with lane l holding A[i = l&15][kk = l>>4], B[kk = l>>4][j = l&15] and D[i = (l>>4) + 4 reg][j = l&15]
// (cdna_hip_programming.md, "f64 MFMA does NOT use the f32 maps").  Here i = OUTPUT column, j = ROW of the panel,
// kk = basis column within a 4-column step, i.e. D^T = C^T X^T.
//   * the A operands (coefficient tiles, one per output group of 16 and k-step) are staged ONCE per block in LDS and read
//     back per MFMA (512 B per wave instruction, lane-indexed: conflict free);
//   * the B operand is this wave's OWN rows of X, 16 bytes per lane, prefetched U k-steps ahead: every wave of the block
//     streams different rows (8 independent streams per CU), and X leaves HBM exactly once;
//   * every wave keeps the accumulators of ALL NG output groups of its rows in registers (NG x 2 row groups independent
//     MFMA chains).
//   real kind   : a lane loads rows (2j, 2j+1) of column 4t+kk and feeds the even rows to one accumulator and the odd rows
//                 to another (32 rows per row group, 16 outputs per group).
//   complex kind: a group is 8 complex outputs as 16 real ones (n < 8: Re, n >= 8: Im).  A lane loads (re, im) of row j of
//                 column 4t+kk; the re parts multiply the tile [Cr | Ci], the im parts the tile [-Ci | Cr], which is the
//                 first tile with lanes n <-> n^8 swapped and one half negated, so only [Cr | Ci] is stored.
// Cp: coefficient tiles packed per lane by pack_coef_mfma: [group][k-step t][64 lanes].
typedef double v4d __attribute__((ext_vector_type(4)));

// PF (round 4; real kind, <= 32 outputs, launched for accumulating calls only: the block Gram-Schmidt's updates Y -= X H): the tile of Y that
// the result is added to is loaded BEFORE the k-loop, 16 more 16-byte loads in flight per lane under the MFMAs, instead of after it, where a
// wave had nothing else to issue: block DGS k = 128, p = 32: 10.3 -> 9.8 ms, k = 32, p = 32: 4.9 -> 4.5 ms.  (Compiled into the plain
// product as well it cost that one 13 %, so it is a template flag; for the complex three-product kernel it changed nothing and is not built.)
template <bool CPLX, int NG, bool PFY = false, bool ROLL = false>
__global__ __launch_bounds__(512) void panel_gemm_mfma(const double *__restrict__ X, int64_t ldx, int k,
                                                       double *__restrict__ Y, int64_t ldy, int qn,
                                                       const double *__restrict__ Cp, int64_t n, int accumulate, int policy, Guard guard) {
    if (stopped(guard)) return;
    constexpr int QB = CPLX ? 8 : 16;            // output columns per group
    constexpr int RG = CPLX ? 16 : 32;           // rows per row group (one MFMA N extent; x2 rows per lane for real)
    constexpr int NACC = CPLX ? 1 : 2;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int U = 4;                         // k-steps of X in flight per wave: 2 U loads of 16 B per lane (8 k-steps: null, tuning log 19; a 124-register variant with
                                                 // U = 2 and TWO blocks per CU measured the same as this one, profiles/r05_ab_gemm_roll.jsonl)
    extern __shared__ double tiles[];            // [NG][nt][64]
    const int nt = (k + 3) >> 2;
    for (int i = threadIdx.x; i < NG * nt * 64; i += blockDim.x) tiles[i] = Cp[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4, j = lane & 15;
    const double sgn = (lane & 8) ? 1.0 : -1.0;  // complex second tile: [-Ci | Cr] from [Cr | Ci]
    const int64_t xstride = ldx * ED, ystride = ldy * ED;
    constexpr int tile_rows = 8 * 2 * RG;        // 8 waves x two row groups
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const double *__restrict__ Xl = X + (int64_t)kk * xstride;      // this lane's column within a k-step
    const bool kfast = (k & 3) == 0;

    constexpr int RING = 4;                      // (eight k-steps for the narrow products, NG <= 2: the same 1.41 / 2.28 ms at k = 64 / 128, q = 32 as four, and as the batch schedule)
    v2d xring[ROLL ? RING : 1][2];               // ROLL, straight-line path: the ring of k-steps of X, alive ACROSS tiles
    bool primed = false;                         // ... and whether it already holds the first U k-steps of the tile about to start
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * tile_rows + (int64_t)wave * (2 * RG) + (CPLX ? j : 2 * j), r1 = r0 + RG;
        const bool fast = kfast && (tile + 1) * tile_rows <= n;
        v4d acc[NG][2][NACC];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int e = 0; e < NACC; ++e) acc[g][g2][e] = v4d{0.0, 0.0, 0.0, 0.0};

        constexpr bool PF = PFY && !CPLX && NG <= 2;
        v2d yin[PF ? NG : 1][2][4];
        if constexpr (PF) {
            if (accumulate) {
                const bool fullp = (tile + 1) * tile_rows <= n;
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int qq = g * QB + kk + 4 * reg;
                            yin[g][g2][reg] = qq < qn ? load_y<CPLX>(Y + (int64_t)qq * ystride, g2 ? r1 : r0, n, fullp) : v2d{0.0, 0.0};
                        }
            }
        }
        // ROLLING prefetch (round 5): a ring of U k-steps of X in registers; the moment a k-step's two 16-byte loads have been handed to
        // the MFMAs, the loads of the k-step U further on take their place -- 2 U loads per lane stay in flight under the MFMAs all
        // the way through the k-loop.  (Rounds 2-4 loaded a BATCH of U k-steps, waited for all of it -- s_waitcnt vmcnt(0) -- and
        // only then issued its 4 U NG MFMAs: every batch paid an HBM round trip with an idle matrix pipe and nothing in flight behind
        // it; a double-buffered variant of that with half-size batches had measured slower.)  ROLL = false keeps the batch schedule.
        if constexpr (ROLL && !CPLX) {
          constexpr int UR = RING;              // ring depth in k-steps: 2 UR loads of 16 B per lane in flight
          if (fast && (nt == 32 || nt == 16)) {
            // straight-line ring (full tile, k = 128 or 64: the launcher sends no other width here): NO branch between the loads and the MFMAs of a k-step, so the compiler keeps
            // exact s_waitcnt vmcnt(2 (UR - 1)) counts instead of draining the queue at every basic-block boundary of the guarded version
            // below.  The ring never runs dry: the refills of the last UR k-steps are the FIRST UR k-steps of this block's next tile (when
            // that one is a full tile too; its own first columns otherwise -- cache hits, never consumed), in flight under the epilogue's stores.
            auto &x = xring;
            // ONE running (scalar) column offset, advanced by a k-step (4 columns) per refill and kept opaque: left to itself the compiler
            // forms the 32 column addresses of the unrolled k-loop ahead of the tile loop and spills them
            const double *__restrict__ xb = Xl + r0;
            const bool next_fast = (tile + gridDim.x + 1) * tile_rows <= n;
            const int64_t dnext = next_fast ? (int64_t)gridDim.x * tile_rows : (int64_t)0;      // (scalar: rows from this tile to the block's next one)
            const int64_t xstep = 4 * xstride;
            if (!primed) {
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    x[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xb + u * xstep));
                    x[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xb + u * xstep + RG));
                }
            }
            primed = next_fast;
            int64_t xo = UR * xstep, xo2 = dnext;
            asm volatile("" : "+s"(xo), "+s"(xo2));
            // one k-step: this step's A operands from LDS, its MFMAs on ring slot u, then the slot's refill -- pinned in that order (the
            // scheduler otherwise hoists the refills of all UR slots to the top of the body and consumes them in the same iteration:
            // the batch schedule again)
            auto kstep = [&](int t, int u, bool tail) {
                double a0[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) a0[g] = tiles[(g * nt + t) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    acc[g][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[g], x[u][0].x, acc[g][0][0], 0, 0, 0);
                    acc[g][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[g], x[u][0].y, acc[g][0][1], 0, 0, 0);
                    acc[g][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[g], x[u][1].x, acc[g][1][0], 0, 0, 0);
                    acc[g][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[g], x[u][1].y, acc[g][1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const double *__restrict__ src = xb + (tail ? xo2 : xo);
                x[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(src));
                x[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(src + RG));
                xo += tail ? (int64_t)0 : xstep;
                xo2 += tail ? xstep : (int64_t)0;
                asm volatile("" : "+s"(xo), "+s"(xo2));
                __builtin_amdgcn_sched_barrier(0);
            };
            // the two basis widths of the restart update and the block Gram-Schmidt (k = 128, 64) fully unrolled: the waitcnt pass drains the
            // whole ring -- s_waitcnt vmcnt(0) -- at the head of a LOOP over the ring (it merges the back edge conservatively), but counts
            // exactly (vmcnt(2 (UR - 1))) along straight-line code
            if (nt == 32) {
#pragma unroll
                for (int t = 0; t < 32; ++t) kstep(t, t & (UR - 1), t + UR >= 32);
            } else if (nt == 16) {                 // (spelled out: the compiler folds the LDS offsets of the unrolled steps only for a known nt)
#pragma unroll
                for (int t = 0; t < 16; ++t) kstep(t, t & (UR - 1), t + UR >= 16);
            }
          } else {
            // (the ragged last tile, or a basis width that is not a multiple of 16: one guarded k-step at a time -- at most one tile per launch takes this path)
            int jl = j, kl = kk;
            asm volatile("" : "+v"(jl), "+v"(kl));                    // (this rare path's addresses formed here, not kept in registers across the unrolled k-loops)
            const int64_t r0s = tile * tile_rows + (int64_t)wave * (2 * RG) + 2 * jl;
            const double *__restrict__ Xs = X + (int64_t)kl * xstride;
#pragma unroll 1
            for (int t = 0; t < nt; ++t) {
                v2d xa = v2d{0.0, 0.0}, xc2 = v2d{0.0, 0.0};
                if ((4 * t + kl) < k) {
                    const double *__restrict__ xc = Xs + (int64_t)(4 * t) * xstride;
                    xa = load_y<CPLX>(xc, r0s, n, false);
                    xc2 = load_y<CPLX>(xc, r0s + RG, n, false);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const double a0 = tiles[(g * nt + t) * 64 + lane];
                    acc[g][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xa.x, acc[g][0][0], 0, 0, 0);
                    acc[g][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xa.y, acc[g][0][1], 0, 0, 0);
                    acc[g][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xc2.x, acc[g][1][0], 0, 0, 0);
                    acc[g][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xc2.y, acc[g][1][1], 0, 0, 0);
                }
            }
          }
        } else if constexpr (ROLL) {
            v2d x[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[u][0] = v2d{0.0, 0.0};
                x[u][1] = v2d{0.0, 0.0};
                if (u < nt) {
                    const double *__restrict__ xc = Xl + (int64_t)(4 * u) * xstride;
                    if (fast) {
                        x[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + r0 * ED));
                        x[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + r1 * ED));
                    } else if ((4 * u + kk) < k) {
                        x[u][0] = load_y<CPLX>(xc, r0, n, false);
                        x[u][1] = load_y<CPLX>(xc, r1, n, false);
                    }
                }
            }
            for (int t0 = 0; t0 < nt; t0 += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u;
                    const v2d xa = x[u][0], xb = x[u][1];
                    x[u][0] = v2d{0.0, 0.0};
                    x[u][1] = v2d{0.0, 0.0};
                    if (t + U < nt) {
                        const double *__restrict__ xc = Xl + (int64_t)(4 * (t + U)) * xstride;
                        if (fast) {
                            x[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + r0 * ED));
                            x[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + r1 * ED));
                        } else if ((4 * (t + U) + kk) < k) {
                            x[u][0] = load_y<CPLX>(xc, r0, n, false);
                            x[u][1] = load_y<CPLX>(xc, r1, n, false);
                        }
                    }
                    if (t < nt) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const double a0 = tiles[(g * nt + t) * 64 + lane];
                            if constexpr (CPLX) {
                                const double a1 = tiles[(g * nt + t) * 64 + (lane ^ 8)] * sgn;
                                acc[g][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xa.x, acc[g][0][0], 0, 0, 0);
                                acc[g][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xb.x, acc[g][1][0], 0, 0, 0);
                                acc[g][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, xa.y, acc[g][0][0], 0, 0, 0);
                                acc[g][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, xb.y, acc[g][1][0], 0, 0, 0);
                            } else {
                                acc[g][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xa.x, acc[g][0][0], 0, 0, 0);
                                acc[g][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xa.y, acc[g][0][1], 0, 0, 0);
                                acc[g][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xb.x, acc[g][1][0], 0, 0, 0);
                                acc[g][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xb.y, acc[g][1][1], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        } else {
        // U k-steps per batch: 2 U loads of 16 B per lane in flight, then their MFMAs.  (A software-pipelined variant with
        // the next batch's loads issued ahead was measured 2-18 % SLOWER: it needs smaller batches to fit the register
        // file, and the two waves per SIMD already overlap each other's load latency.)
        for (int t0 = 0; t0 < nt; t0 += U) {
            v2d x[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u;
                x[u][0] = v2d{0.0, 0.0};
                x[u][1] = v2d{0.0, 0.0};
                if (t < nt) {
                    const double *__restrict__ xc = Xl + (int64_t)(4 * t) * xstride;
                    if (fast) {
                        x[u][0] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + r0 * ED));
                        x[u][1] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + r1 * ED));
                    } else if ((4 * t + kk) < k) {
                        x[u][0] = load_y<CPLX>(xc, r0, n, false);
                        x[u][1] = load_y<CPLX>(xc, r1, n, false);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u;
                if (t < nt) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double a0 = tiles[(g * nt + t) * 64 + lane];
                        if constexpr (CPLX) {
                            const double a1 = tiles[(g * nt + t) * 64 + (lane ^ 8)] * sgn;
#pragma unroll
                            for (int g2 = 0; g2 < 2; ++g2)
                                acc[g][g2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x[u][g2].x, acc[g][g2][0], 0, 0, 0);
#pragma unroll
                            for (int g2 = 0; g2 < 2; ++g2)
                                acc[g][g2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x[u][g2].y, acc[g][g2][0], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int g2 = 0; g2 < 2; ++g2) {
                                acc[g][g2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x[u][g2].x, acc[g][g2][0], 0, 0, 0);
                                acc[g][g2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x[u][g2].y, acc[g][g2][1], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        }
        // D[n = kk + 4 reg][row j] of group g
        const bool full = (tile + 1) * tile_rows <= n;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int64_t r = g2 ? r1 : r0;
                if constexpr (CPLX) {
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {              // complex output kk + 4 h2: (Re, Im) = regs (h2, h2 + 2)
                        const int qq = g * QB + kk + 4 * h2;
                        if (qq < qn) {
                            double *yc = Y + (int64_t)qq * ystride;
                            v2d out = v2d{acc[g][g2][0][h2], acc[g][g2][0][h2 + 2]};
                            if (accumulate) out += load_y<CPLX>(yc, r, n, full);
                            store_rows<CPLX>(yc, r, n, full, out, policy);
                        }
                    }
                } else {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {           // real output kk + 4 reg: rows (2j, 2j+1) = (even, odd acc)
                        const int qq = g * QB + kk + 4 * reg;
                        if (qq < qn) {
                            double *yc = Y + (int64_t)qq * ystride;
                            v2d out = v2d{acc[g][g2][0][reg], acc[g][g2][1][reg]};
                            if constexpr (PF) { if (accumulate) out += yin[g][g2][reg]; }
                            else if (accumulate) out += load_y<CPLX>(yc, r, n, full);
                            store_rows<CPLX>(yc, r, n, full, out, policy);
                        }
                    }
                }
            }
        }
    }
}

// Complex tall-skinny product with THREE real products per complex one (round 4).  panel_gemm_mfma<true> computes a complex
// product as a real one of doubled size -- four real multiplications per complex multiplication, 8 flop -- and runs AT the
// rate the batch schedule reaches (47-50 TFLOP/s of the 77.6 the instruction sustains, tools/mfma_f64_peak.hip); fewer flops is the other way down:
//     P1 = Xr Cr,   P2 = Xi Ci,   P3 = (Xr + Xi)(Cr + Ci)      =>      Re(X C) = P1 - P2,   Im(X C) = P3 - P1 - P2
// (Karatsuba / "3M": 6 flop per complex multiplication).  A group is 16 COMPLEX outputs (all 16 rows of the A operand); per
// k-step of four basis columns and row group three MFMAs -- A = Cr | Ci | Cr + Ci, B = xr | xi | xr + xi of the lane's row -- into
// three accumulators, combined when the tile is stored.  Only Cr and Ci are staged in LDS (the sum is one add per k-step and group).
// Rounding: Im(X C) carries the cancellation of P3 - P1 - P2, i.e. an error of order eps (|Xr| + |Xi|)(|Cr| + |Ci|) per term --
// bounded NORMWISE like the 4-multiplication form (the comparisons of this suite are normwise), not componentwise.
// Cp: [group][k-step][Cr | Ci][64 lanes] from pack_coef_mfma3m.
// ROLL (round 5): the k-steps of X as a ring refilled step by step on straight-line code and carried across tiles, as panel_gemm_mfma's
// (full tiles, k = 128 or 64); the batch schedule otherwise.
template <int NG, int NR, bool ROLL = false>  // NR: row groups of 16 rows per wave (2; 1 for 64 outputs, whose 12 accumulators of 8 registers leave no room for 24)
__global__ __launch_bounds__(512) void panel_gemm_mfma3m(const double *__restrict__ X, int64_t ldx, int k,
                                                         double *__restrict__ Y, int64_t ldy, int qn,
                                                         const double *__restrict__ Cp, int64_t n, int accumulate, int policy, Guard guard) {
    if (stopped(guard)) return;
    constexpr int QB = 16;                       // complex output columns per group
    constexpr int RG = 16;                       // rows per row group (the MFMA's N extent)
    constexpr int U = 8 / NR;                    // k-steps of X in flight per wave: NR U = 8 loads of 16 B per lane
    extern __shared__ double tiles3[];           // [NG][nt][2][64]
    const int nt = (k + 3) >> 2;
    for (int i = threadIdx.x; i < NG * nt * 128; i += blockDim.x) tiles3[i] = Cp[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kk = lane >> 4, j = lane & 15;
    const int64_t xstride = ldx * 2, ystride = ldy * 2;
    constexpr int tile_rows = 8 * NR * RG;       // 8 waves x NR row groups
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const double *__restrict__ Xl = X + (int64_t)kk * xstride;      // this lane's column within a k-step
    const bool kfast = (k & 3) == 0;

    constexpr int UR = 4;                        // ring depth in k-steps
    v2d xring[ROLL ? UR : 1][NR];                // ROLL: the ring of k-steps of X, alive ACROSS tiles
    bool primed = false;                         // ... and whether it already holds the first U k-steps of the tile about to start
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * tile_rows + (int64_t)wave * (NR * RG) + j;
        const bool fast = kfast && (tile + 1) * tile_rows <= n;
        v4d acc[NG][NR][3];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int g2 = 0; g2 < NR; ++g2)
#pragma unroll
                for (int e = 0; e < 3; ++e) acc[g][g2][e] = v4d{0.0, 0.0, 0.0, 0.0};

        bool done = false;
        int64_t xs = xstride;
        if constexpr (ROLL) asm volatile("" : "+s"(xs));      // (the lane's column base re-formed per tile: kept across the unrolled k-loop it was spilled)
        const double *__restrict__ Xt = X + (int64_t)kk * xs;
        if constexpr (ROLL) {
            if (fast && (nt == 32 || nt == 16)) {
                // (see panel_gemm_mfma: one running scalar column offset kept opaque, sched_barriers pinning `operands | MFMAs | refill` per
                // k-step, the k-loop fully unrolled for k = 128 / 64 so that the waitcnt pass counts exactly, and the refills of the last UR
                // k-steps being the first UR k-steps of this block's next tile)
                auto &x = xring;
                const double *__restrict__ xb = Xt + r0 * 2;
                const bool next_fast = (tile + gridDim.x + 1) * tile_rows <= n;
                const int64_t dnext = next_fast ? (int64_t)gridDim.x * tile_rows * 2 : (int64_t)0;
                const int64_t xstep = 4 * xstride;
                if (!primed) {
#pragma unroll
                    for (int u = 0; u < UR; ++u)
#pragma unroll
                        for (int g2 = 0; g2 < NR; ++g2)
                            x[u][g2] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xb + u * xstep + g2 * RG * 2));
                }
                primed = next_fast;
                int64_t xo = UR * xstep, xo2 = dnext;
                asm volatile("" : "+s"(xo), "+s"(xo2));
                auto kstep = [&](int t, int u, bool tail) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double ar = tiles3[((g * nt + t) * 2 + 0) * 64 + lane];
                        const double ai = tiles3[((g * nt + t) * 2 + 1) * 64 + lane];
                        const double as = ar + ai;
#pragma unroll
                        for (int g2 = 0; g2 < NR; ++g2) {
                            acc[g][g2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, x[u][g2].x, acc[g][g2][0], 0, 0, 0);
                            acc[g][g2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, x[u][g2].y, acc[g][g2][1], 0, 0, 0);
                            acc[g][g2][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, x[u][g2].x + x[u][g2].y, acc[g][g2][2], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const double *__restrict__ src = xb + (tail ? xo2 : xo);
#pragma unroll
                    for (int g2 = 0; g2 < NR; ++g2) x[u][g2] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(src + g2 * RG * 2));
                    xo += tail ? (int64_t)0 : xstep;
                    xo2 += tail ? xstep : (int64_t)0;
                    asm volatile("" : "+s"(xo), "+s"(xo2));
                    __builtin_amdgcn_sched_barrier(0);
                };
                if (nt == 32) {
#pragma unroll
                    for (int t = 0; t < 32; ++t) kstep(t, t & (UR - 1), t + UR >= 32);
                } else if (nt == 16) {             // (spelled out: the compiler folds the LDS offsets of the unrolled steps only for a known nt)
#pragma unroll
                    for (int t = 0; t < 16; ++t) kstep(t, t & (UR - 1), t + UR >= 16);
                }
                done = true;
            }
        }
        if constexpr (ROLL) {
            // (the ragged last tile: one guarded k-step at a time -- at most one tile per launch)
            if (!done) {
                int jl = j, kl = kk;
                asm volatile("" : "+v"(jl), "+v"(kl));                // (this rare path's addresses formed here, not kept in registers across the unrolled k-loops)
                const int64_t r0s = tile * tile_rows + (int64_t)wave * (NR * RG) + jl;
                const double *__restrict__ Xs = X + (int64_t)kl * xs;
#pragma unroll 1
                for (int t = 0; t < nt; ++t) {
                    v2d xg[NR];
#pragma unroll
                    for (int g2 = 0; g2 < NR; ++g2) {
                        xg[g2] = v2d{0.0, 0.0};
                        if ((4 * t + kl) < k) xg[g2] = load_y<true>(Xs + (int64_t)(4 * t) * xstride, r0s + g2 * RG, n, false);
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double ar = tiles3[((g * nt + t) * 2 + 0) * 64 + lane];
                        const double ai = tiles3[((g * nt + t) * 2 + 1) * 64 + lane];
                        const double as = ar + ai;
#pragma unroll
                        for (int g2 = 0; g2 < NR; ++g2) {
                            acc[g][g2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, xg[g2].x, acc[g][g2][0], 0, 0, 0);
                            acc[g][g2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, xg[g2].y, acc[g][g2][1], 0, 0, 0);
                            acc[g][g2][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, xg[g2].x + xg[g2].y, acc[g][g2][2], 0, 0, 0);
                        }
                    }
                }
            }
        } else
        for (int t0 = 0; t0 < nt; t0 += U) {
            v2d x[U][NR];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u;
#pragma unroll
                for (int g2 = 0; g2 < NR; ++g2) {
                    x[u][g2] = v2d{0.0, 0.0};
                    if (t < nt) {
                        const double *__restrict__ xc = Xl + (int64_t)(4 * t) * xstride;
                        if (fast) x[u][g2] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(xc + (r0 + g2 * RG) * 2));
                        else if ((4 * t + kk) < k) x[u][g2] = load_y<true>(xc, r0 + g2 * RG, n, false);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u;
                if (t < nt) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double ar = tiles3[((g * nt + t) * 2 + 0) * 64 + lane];
                        const double ai = tiles3[((g * nt + t) * 2 + 1) * 64 + lane];
                        const double as = ar + ai;
#pragma unroll
                        for (int g2 = 0; g2 < NR; ++g2) {
                            acc[g][g2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, x[u][g2].x, acc[g][g2][0], 0, 0, 0);
                            acc[g][g2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, x[u][g2].y, acc[g][g2][1], 0, 0, 0);
                            acc[g][g2][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, x[u][g2].x + x[u][g2].y, acc[g][g2][2], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // D[output i = kk + 4 reg][row j] of group g: (Re, Im) = (P1 - P2, P3 - P1 - P2)
        const bool full = (tile + 1) * tile_rows <= n;
        int64_t ys = ystride;
        if constexpr (ROLL) asm volatile("" : "+s"(ys));      // (keeps the 4 NG column addresses of Y out of the registers the unrolled k-loop needs: they were hoisted and spilled)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int g2 = 0; g2 < NR; ++g2) {
                const int64_t r = r0 + g2 * RG;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int qq = g * QB + kk + 4 * reg;
                    if (qq < qn) {
                        double *yc = Y + (int64_t)qq * ys;
                        const double p1 = acc[g][g2][0][reg], p2 = acc[g][g2][1][reg], p3 = acc[g][g2][2][reg];
                        v2d out = v2d{p1 - p2, (p3 - p1) - p2};
                        if (accumulate) out += load_y<true>(yc, r, n, full);
                        store_rows<true>(yc, r, n, full, out, policy);
                    }
                }
            }
        }
    }
}

// coefficient tiles of panel_gemm_mfma3m from device coefficients laid out [q][ldc][2] (column-major k x q, complex):
// Cp[((g * nt + t) * 2 + part) * 64 + lane] = sign * (part ? Im : Re) C(col = 4 t + (lane >> 4), q = 16 g + (lane & 15)).
__global__ __launch_bounds__(256) void pack_coef_mfma3m(const double *__restrict__ C, int64_t ldc, int k, int q, double sign,
                                                        double *__restrict__ Cp) {
    const int ngroups = (q + 15) / 16, nt = (k + 3) >> 2;
    const int total = ngroups * nt * 128;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, part = (idx >> 6) & 1, t = (idx >> 7) % nt, g = idx / (128 * nt);
        const int col = 4 * t + (lane >> 4), qq = g * 16 + (lane & 15);
        Cp[idx] = (col < k && qq < q) ? sign * C[((int64_t)qq * ldc + col) * 2 + part] : 0.0;
    }
}

// M = X(:, :k)^H Y(:, :p) on the FP64 matrix cores: ONE pass over X (and Y) for up to 128 x 128 results -- Gram matrices,
// innerprod_matrix and the coefficient passes of the block Gram-Schmidt with many right-hand sides, where panel_dot_p
// (VALU, <= 4 right-hand sides per pass) would read X p/4 times.
//   D(16x16) += A(16x4) B(4x16) with the contraction index on the ROWS of the panel: A[i][kk] = X(r + kk, 16 I + i),
//   B[kk][j] = Y(r + kk, 16 J + j).  Lanes of an MFMA operand run along COLUMNS of the panel, the opposite of how the panel
//   lies in memory, so a block stages a tile of 64 real rows x all columns in LDS -- 16-byte coalesced loads along the rows,
//   row stride 66 doubles per column, which makes the lane-indexed ds_read_b64 of an operand (16 columns x 4 rows) bank-
//   conflict free -- and prefetches the next tile into registers while the MFMAs of the current one run.
//   Wave w owns tile row I = w (16 columns of X) and keeps the accumulators of all J tiles (<= 8) of that row in registers;
//   for k <= 64 the 8 waves split NI ways over tile rows and 8/NI ways over the k-steps of a tile.
//   complex kind: the panel is read as a REAL one of 2n rows (re, im interleaved): Re M = Xr^T Yr, and
//   Im M = Xr^T Y~ with Y~(2r) = Yi(r), Y~(2r+1) = -Yr(r), i.e. the B operand read one row over with a sign: two MFMAs per
//   tile and k-step, conj on X as in dotc.
//   flags: 1 = Y is X (Gram: one tile serves both operands, no norm slots), 2 = upper tiles only (J >= I; tile rows are
//   dealt so that every SIMD gets 9 of the 36 tiles).
//   Results: partial[vb][slot], vb = block * (8/NI) + row group, slot = (q (k+1) + i) * ED + part -- panel_dot_p's layout
//   with the norms ||Y_q||^2 (npartial[block][q]) in slot i = k -- summed over vb in fixed order by finish_xhy.
// PJM = J tiles a wave can hold (8: up to 128 right-hand sides; 2: up to 32, half the registers), TR = real rows per tile
// (64, or 32 for the small variant: 40 KB of LDS at k = 128, so that two or three blocks share a CU and cover each other's
// barriers when there are only a few MFMAs per tile).
template <bool CPLX, int PJM, int TR, bool DB = false>
__global__ __launch_bounds__(512) void panel_xhy_mfma(const double *__restrict__ X, int64_t ldx, int k,
                                                      const double *__restrict__ Y, int64_t ldy, int p, int64_t n, int flags,
                                                      int NI, double *__restrict__ partial, double *__restrict__ npartial) {
    constexpr int ER = CPLX ? 2 : 1;
    constexpr int S = TR + 2;                    // LDS row stride of a column (S mod 32 == 2: conflict-free operand reads)
    constexpr int CH = TR / 2, CHS = TR == 64 ? 5 : 4;   // 16-byte chunks per column of a tile (and its log2)
    constexpr int CPP = 512 / CH;                // columns one block-wide pass stages
    constexpr int NXP = 128 / CPP, NYP = (16 * PJM + CPP - 1) / CPP;
    static_assert(TR == 64 || TR == 32, "tile rows");
    extern __shared__ __attribute__((aligned(16))) double xhy_lds[];   // 16-byte ds_write_b128 of the staged chunks
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int KP = (k + 15) >> 4, PJ = (p + 15) >> 4;
    const int KS = (k + CPP - 1) / CPP, PS = (p + CPP - 1) / CPP;     // staging passes
    const bool alias = flags & 1, upper = flags & 2;
    double *Xt = xhy_lds, *Yt = alias ? xhy_lds : xhy_lds + KP * 16 * S;
    const int64_t nr = n * ER;
    const int64_t ntiles = (nr + TR - 1) / TR;
    const int64_t xcs = ldx * ER, ycs = ldy * ER;
    const int WR = 8 / NI;
    int wi = wave % NI;
    const int wr = wave / NI;
    if (upper && NI == 8 && wave >= 4) wi = 11 - wave;
    const bool active = wi < KP;
    const int arow = lane >> 4, acol = lane & 15;

    v4d acc_re[PJM], acc_im[PJM];
#pragma unroll
    for (int J = 0; J < PJM; ++J) { acc_re[J] = v4d{0.0, 0.0, 0.0, 0.0}; acc_im[J] = v4d{0.0, 0.0, 0.0, 0.0}; }
    double nacc[NYP];
#pragma unroll
    for (int s = 0; s < NYP; ++s) nacc[s] = 0.0;
    v2d xs[NXP], ys[NYP];

#ifdef LK_DIAGNOSTICS
    const bool dbg_nomfma = flags & 16, dbg_noload = flags & 32;      // diagnostics build only (tools/bench_gram.py xhy_debug=...): WRONG results, phase timing
#else
    constexpr bool dbg_nomfma = false, dbg_noload = false;
#endif
    auto gload = [&](int64_t T) {
        if (dbg_noload && T != (int64_t)blockIdx.x) return;
        const int64_t rbase = T * TR;
#pragma unroll
        for (int s = 0; s < NXP; ++s) {
            xs[s] = v2d{0.0, 0.0};
            if (s < KS) {
                const int c = t + 512 * s, col = c >> CHS;
                const int64_t rr = rbase + 2 * (c & (CH - 1));
                if (col < k) {
                    const double *pc = X + (int64_t)col * xcs;
                    if (rr + 1 < nr) xs[s] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(pc + rr));
                    else if (rr < nr) xs[s].x = pc[rr];
                }
            }
        }
        if (!alias) {
#pragma unroll
            for (int s = 0; s < NYP; ++s) {
                ys[s] = v2d{0.0, 0.0};
                if (s < PS) {
                    const int c = t + 512 * s, col = c >> CHS;
                    const int64_t rr = rbase + 2 * (c & (CH - 1));
                    if (col < p) {
                        const double *pc = Y + (int64_t)col * ycs;
                        if (rr + 1 < nr) ys[s] = *reinterpret_cast<const v2d *>(pc + rr);
                        else if (rr < nr) ys[s].x = pc[rr];
                    }
                }
            }
        }
    };

    // staged chunks -> LDS tile `Xb` / `Yb` (and the norms of Y, once per tile)
    auto stage = [&](double *Xb, double *Yb) {
#pragma unroll
        for (int s = 0; s < NXP; ++s)
            if (s < KS) {
                const int c = t + 512 * s;
                if ((c >> CHS) < KP * 16) *reinterpret_cast<v2d *>(Xb + (c >> CHS) * S + 2 * (c & (CH - 1))) = xs[s];
            }
        if (!alias) {
#pragma unroll
            for (int s = 0; s < NYP; ++s)
                if (s < PS) {
                    const int c = t + 512 * s;
                    if ((c >> CHS) < PJ * 16) *reinterpret_cast<v2d *>(Yb + (c >> CHS) * S + 2 * (c & (CH - 1))) = ys[s];
                    nacc[s] += ys[s].x * ys[s].x + ys[s].y * ys[s].y;
                }
        }
    };
    // this wave's MFMAs on the tile staged in `Xb` / `Yb`
    auto contract = [&](const double *Xb, const double *Yb) {
        if (dbg_nomfma) return;
        if (active) {
            for (int step = wr; step < TR / 4; step += WR) {
                const int ro = 4 * step + arow;
                const double a = Xb[(16 * wi + acol) * S + ro];
#pragma unroll
                for (int J = 0; J < PJM; ++J) {
                    if (J < PJ && (!upper || J >= wi)) {
                        const double b = Yb[(16 * J + acol) * S + ro];
                        acc_re[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc_re[J], 0, 0, 0);
                        if constexpr (CPLX) {
                            double b2 = Yb[(16 * J + acol) * S + (ro ^ 1)];
                            b2 = (ro & 1) ? -b2 : b2;
                            acc_im[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc_im[J], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };

    int64_t T = blockIdx.x;
    if constexpr (DB) {
        // DOUBLE-BUFFERED tile (round 5): ONE barrier per tile.  While the MFMAs of tile T run out of one buffer, every wave -- as it
        // finishes its own share -- stages tile T + grid (in its registers since the previous iteration) into the other buffer and
        // sends the loads of tile T + 2 grid on their way; a light wave does that under a heavy wave's MFMAs (the two share a SIMD),
        // and the matrix pipe no longer idles through a second barrier and a block-wide staging phase per tile.
        const int BUF = (KP + (alias ? 0 : PJ)) * 16 * S;            // doubles per buffer
        if (T < ntiles) { gload(T); stage(Xt, Yt); }
        if (T + gridDim.x < ntiles) gload(T + gridDim.x);
        __syncthreads();
        int buf = 0;
        for (; T < ntiles; T += gridDim.x, buf ^= 1) {
            contract(Xt + buf * BUF, Yt + buf * BUF);
            if (T + gridDim.x < ntiles) stage(Xt + (buf ^ 1) * BUF, Yt + (buf ^ 1) * BUF);
            if (T + 2 * (int64_t)gridDim.x < ntiles) gload(T + 2 * (int64_t)gridDim.x);
            __syncthreads();                                        // buffer `buf` has been read by all, buffer `buf ^ 1` is complete
        }
    } else {
        if (T < ntiles) gload(T);
        for (; T < ntiles; T += gridDim.x) {
            __syncthreads();                                        // the previous tile's operands have been read
            stage(Xt, Yt);
            __syncthreads();
            if (T + gridDim.x < ntiles) gload(T + gridDim.x);       // in flight while this tile's MFMAs run
            contract(Xt, Yt);
        }
    }

    const int64_t nslots = (int64_t)p * (k + 1) * ER;
    double *pb = partial + ((int64_t)blockIdx.x * WR + wr) * nslots;
    if (active) {
#pragma unroll
        for (int J = 0; J < PJM; ++J) {
            if (J < PJ && (!upper || J >= wi)) {
                const int q = 16 * J + acol;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * wi + arow + 4 * r;
                    if (i < k && q < p) {
                        pb[((int64_t)q * (k + 1) + i) * ER] = acc_re[J][r];
                        if constexpr (CPLX) pb[((int64_t)q * (k + 1) + i) * ER + 1] = acc_im[J][r];
                    }
                }
            }
        }
    }
    if (!alias) {
#pragma unroll
        for (int s = 0; s < NYP; ++s) {
            if (s < PS) {
                double v = nacc[s];
                if constexpr (CH == 32) v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                const int col = (t >> CHS) + CPP * s;
                if ((t & (CH - 1)) == 0 && col < p) npartial[(int64_t)blockIdx.x * p + col] = v;
            }
        }
    }
}

// X^H Y for the complex kind with THREE real products per complex one (round 4; <= 32 right-hand sides, the shape of the block
// Gram-Schmidt's coefficient passes and of innerprod_matrix).  panel_xhy_mfma<true> reads the panel as a real one of 2n rows and spends
// two MFMAs per k-step of two complex rows (4 real multiplications per complex one); here a k-step is FOUR complex rows and
//     P1 = Xr^T Yr,   P2 = Xi^T Yi,   P3 = (Xr + Xi)^T (Yi - Yr)      =>      Re M = P1 + P2,   Im M = P3 + P1 - P2      (conj on X)
// -- three MFMAs.  The real and imaginary parts are staged as SEPARATE planes (two 8-byte LDS writes per 16-byte load; an operand read
// with stride 2 in an interleaved tile cannot be made bank-conflict free), each [columns][S = 18]: 16 complex rows per tile, column
// stride 18 words (18 c mod 32 runs through the even residues: conflict-free operand reads).  46 KB of LDS at k = 128, p = 32: three
// blocks per CU.  Same grid, wave roles, partial layout and norms as panel_xhy_mfma<true, 2, 32>; no aliasing (Gram keeps the 4-product form).
__global__ __launch_bounds__(512) void panel_xhy_mfma3m(const double *__restrict__ X, int64_t ldx, int k,
                                                        const double *__restrict__ Y, int64_t ldy, int p, int64_t n, int flags,
                                                        int NI, double *__restrict__ partial, double *__restrict__ npartial) {
    constexpr int PJM = 2, TRC = 16, S = TRC + 2, CHS = 4, CPP = 512 / TRC;      // 32 columns staged per block-wide pass
    constexpr int NXP = 128 / CPP, NYP = (16 * PJM + CPP - 1) / CPP;
    extern __shared__ __attribute__((aligned(16))) double xh3_lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int KP = (k + 15) >> 4, PJ = (p + 15) >> 4;
    const int KS = (k + CPP - 1) / CPP, PS = (p + CPP - 1) / CPP;
    const bool upper = flags & 2;
    double *Xr = xh3_lds, *Xi = Xr + KP * 16 * S, *Yr = Xi + KP * 16 * S, *Yi = Yr + PJ * 16 * S;
    const int64_t ntiles = (n + TRC - 1) / TRC;
    const int64_t xcs = ldx * 2, ycs = ldy * 2;
    const int WR = 8 / NI;
    const int wi = wave % NI, wr = wave / NI;
    const bool active = wi < KP;
    const int arow = lane >> 4, acol = lane & 15;

    v4d p1[PJM], p2[PJM], p3[PJM];
#pragma unroll
    for (int J = 0; J < PJM; ++J) { p1[J] = v4d{0.0, 0.0, 0.0, 0.0}; p2[J] = v4d{0.0, 0.0, 0.0, 0.0}; p3[J] = v4d{0.0, 0.0, 0.0, 0.0}; }
    double nacc[NYP];
#pragma unroll
    for (int s = 0; s < NYP; ++s) nacc[s] = 0.0;
    v2d xs[NXP], ys[NYP];

    auto gload = [&](int64_t T) {
        const int64_t rbase = T * TRC;
#pragma unroll
        for (int s = 0; s < NXP; ++s) {
            xs[s] = v2d{0.0, 0.0};
            if (s < KS) {
                const int c = t + 512 * s, col = c >> CHS;
                const int64_t rr = rbase + (c & (TRC - 1));
                if (col < k && rr < n) xs[s] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(X + (int64_t)col * xcs + 2 * rr));
            }
        }
#pragma unroll
        for (int s = 0; s < NYP; ++s) {
            ys[s] = v2d{0.0, 0.0};
            if (s < PS) {
                const int c = t + 512 * s, col = c >> CHS;
                const int64_t rr = rbase + (c & (TRC - 1));
                if (col < p && rr < n) ys[s] = *reinterpret_cast<const v2d *>(Y + (int64_t)col * ycs + 2 * rr);
            }
        }
    };

    int64_t T = blockIdx.x;
    if (T < ntiles) gload(T);
    for (; T < ntiles; T += gridDim.x) {
        __syncthreads();                                            // the previous tile's operands have been read
#pragma unroll
        for (int s = 0; s < NXP; ++s)
            if (s < KS) {
                const int c = t + 512 * s;
                if ((c >> CHS) < KP * 16) {
                    Xr[(c >> CHS) * S + (c & (TRC - 1))] = xs[s].x;
                    Xi[(c >> CHS) * S + (c & (TRC - 1))] = xs[s].y;
                }
            }
#pragma unroll
        for (int s = 0; s < NYP; ++s)
            if (s < PS) {
                const int c = t + 512 * s;
                if ((c >> CHS) < PJ * 16) {
                    Yr[(c >> CHS) * S + (c & (TRC - 1))] = ys[s].x;
                    Yi[(c >> CHS) * S + (c & (TRC - 1))] = ys[s].y;
                }
                nacc[s] += ys[s].x * ys[s].x + ys[s].y * ys[s].y;
            }
        __syncthreads();
        if (T + gridDim.x < ntiles) gload(T + gridDim.x);           // in flight while this tile's MFMAs run
        if (active) {
            for (int step = wr; step < TRC / 4; step += WR) {
                const int ro = 4 * step + arow;
                const double ar = Xr[(16 * wi + acol) * S + ro], ai = Xi[(16 * wi + acol) * S + ro];
                const double as = ar + ai;
#pragma unroll
                for (int J = 0; J < PJM; ++J) {
                    if (J < PJ && (!upper || J >= wi)) {
                        const double br = Yr[(16 * J + acol) * S + ro], bi = Yi[(16 * J + acol) * S + ro];
                        p1[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, p1[J], 0, 0, 0);
                        p2[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, p2[J], 0, 0, 0);
                        p3[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, bi - br, p3[J], 0, 0, 0);
                    }
                }
            }
        }
    }

    const int64_t nslots = (int64_t)p * (k + 1) * 2;
    double *pb = partial + ((int64_t)blockIdx.x * WR + wr) * nslots;
    if (active) {
#pragma unroll
        for (int J = 0; J < PJM; ++J) {
            if (J < PJ && (!upper || J >= wi)) {
                const int q = 16 * J + acol;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * wi + arow + 4 * r;
                    if (i < k && q < p) {
                        pb[((int64_t)q * (k + 1) + i) * 2] = p1[J][r] + p2[J][r];
                        pb[((int64_t)q * (k + 1) + i) * 2 + 1] = (p3[J][r] + p1[J][r]) - p2[J][r];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NYP; ++s) {
        if (s < PS) {
            double v = nacc[s];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            const int col = (t >> CHS) + CPP * s;
            if ((t & (TRC - 1)) == 0 && col < p) npartial[(int64_t)blockIdx.x * p + col] = v;
        }
    }
}

// Gram matrix G = X^H X of a complex basis (k <= 128), upper tiles only, with three real products per complex one (round 4):
//     P1 = Xr_I^T Xr_J,   P2 = Xi_I^T Xi_J,   P3 = (Xr + Xi)_I^T (Xi - Xr)_J      =>      Re G_IJ = P1 + P2,   Im G_IJ = P3 + P1 - P2.
// panel_xhy_mfma<true, 8, 64> gives every wave one tile ROW (up to 8 tiles, two accumulators each: 128 registers); three accumulators
// per tile do not fit that way, so here the KP (KP + 1) / 2 upper tiles (I <= J) are DEALT to the 8 waves round-robin -- 36 tiles at
// k = 128: five for waves 0-3, four for waves 4-7, i.e. nine per SIMD (waves w and w + 4 share one) -- and a wave reads the A operand of
// each of its tiles itself (four 8-byte LDS reads per three MFMAs: ~8 % of the LDS rate).  Real and imaginary parts are staged as separate
// planes, 32 complex rows per tile, column stride 34 words (conflict-free operand reads); the next tile's loads are in flight while the
// current tile's MFMAs run.  Results: partial[block][slot], slot = (j (k + 1) + i) * 2 (+1) for i in tile row I, j in tile column J >= I --
// panel_xhy_mfma's layout with flags = 3 (Y is X, upper tiles only), summed by finish_xhy; no norm slots.
__global__ __launch_bounds__(512) void panel_gram_mfma3m(const double *__restrict__ X, int64_t ldx, int k, int64_t n,
                                                         double *__restrict__ partial) {
    constexpr int TRC = 32, S = TRC + 2, CHS = 5, CPP = 512 / TRC, NXP = 128 / CPP;      // 16 columns staged per block-wide pass
    constexpr int MAXT = 5;                                                             // ceil(36 / 8) tiles per wave
    extern __shared__ __attribute__((aligned(16))) double gr3_lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int KP = (k + 15) >> 4, KS = (k + CPP - 1) / CPP;
    double *Xr = gr3_lds, *Xi = Xr + KP * 16 * S;
    const int64_t ntiles = (n + TRC - 1) / TRC;
    const int64_t xcs = ldx * 2;
    const int arow = lane >> 4, acol = lane & 15;

    // this wave's tiles: upper tiles in the order (0,0), (0,1), ..., (0,KP-1), (1,1), ...; tile number == wave (mod 8)
    int tI[MAXT], tJ[MAXT], nt = 0;
    {
        int idx = 0;
        for (int I = 0; I < KP; ++I)
            for (int J = I; J < KP; ++J, ++idx)
                if ((idx & 7) == wave && nt < MAXT) { tI[nt] = I; tJ[nt] = J; ++nt; }
        for (int q = nt; q < MAXT; ++q) { tI[q] = 0; tJ[q] = 0; }
    }

    v4d p1[MAXT], p2[MAXT], p3[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) { p1[q] = v4d{0.0, 0.0, 0.0, 0.0}; p2[q] = v4d{0.0, 0.0, 0.0, 0.0}; p3[q] = v4d{0.0, 0.0, 0.0, 0.0}; }
    v2d xs[NXP];

    auto gload = [&](int64_t T) {
        const int64_t rbase = T * TRC;
#pragma unroll
        for (int s = 0; s < NXP; ++s) {
            xs[s] = v2d{0.0, 0.0};
            if (s < KS) {
                const int c = t + 512 * s, col = c >> CHS;
                const int64_t rr = rbase + (c & (TRC - 1));
                if (col < k && rr < n) xs[s] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(X + (int64_t)col * xcs + 2 * rr));
            }
        }
    };

    int64_t T = blockIdx.x;
    if (T < ntiles) gload(T);
    for (; T < ntiles; T += gridDim.x) {
        __syncthreads();                                            // the previous tile's operands have been read
#pragma unroll
        for (int s = 0; s < NXP; ++s)
            if (s < KS) {
                const int c = t + 512 * s;
                if ((c >> CHS) < KP * 16) {
                    Xr[(c >> CHS) * S + (c & (TRC - 1))] = xs[s].x;
                    Xi[(c >> CHS) * S + (c & (TRC - 1))] = xs[s].y;
                }
            }
        __syncthreads();
        if (T + gridDim.x < ntiles) gload(T + gridDim.x);           // in flight while this tile's MFMAs run
        for (int step = 0; step < TRC / 4; ++step) {
            const int ro = 4 * step + arow;
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                if (q < nt) {
                    const double ar = Xr[(16 * tI[q] + acol) * S + ro], ai = Xi[(16 * tI[q] + acol) * S + ro];
                    const double br = Xr[(16 * tJ[q] + acol) * S + ro], bi = Xi[(16 * tJ[q] + acol) * S + ro];
                    p1[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, p1[q], 0, 0, 0);
                    p2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, p2[q], 0, 0, 0);
                    p3[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar + ai, bi - br, p3[q], 0, 0, 0);
                }
            }
        }
    }

    const int64_t nslots = (int64_t)k * (k + 1) * 2;
    double *pb = partial + (int64_t)blockIdx.x * nslots;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        if (q < nt) {
            const int j = 16 * tJ[q] + acol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * tI[q] + arow + 4 * r;
                if (i < k && j < k) {
                    pb[((int64_t)j * (k + 1) + i) * 2] = p1[q][r] + p2[q][r];
                    pb[((int64_t)j * (k + 1) + i) * 2 + 1] = (p3[q][r] + p1[q][r]) - p2[q][r];
                }
            }
        }
    }
}

// Gram matrix G = X^T X of a REAL basis of 33..128 columns: ROW-SPLIT deal, tiles staged by LDS-DMA (round 6; gram_matrix, AbstractVectors.fypp:645-657).
// The cyclic deals above give a wave a few tiles on every row step -- one LDS operand read per MFMA -- and only 4 and 8 column blocks divide the eight waves; their
// staging goes through registers (two sets of loads in flight per thread, written to LDS by the thread).  Here
//  * the ROWS of the staged 32-row tile are dealt: wave w takes the row steps 2 (w & 3), 2 (w & 3) + 1 (eight rows) for HALF of the upper-tile list (waves 0-3 the first
//    half, waves 4-7 the second: waves w and w + 4 share a SIMD, so every SIMD runs the same number of MFMAs whatever the split).  A wave reads each of its column
//    blocks ONCE per row step and feeds all its tiles from those registers -- KP reads per ceil(KP (KP + 1) / 4) MFMAs (k = 96: 6 per 10.5 instead of 10 per 9) -- in
//    straight-line code (the tile list is a compile-time constant) for ANY number of column blocks KP = 3..8; accumulators: 8 ceil(KP (KP + 1) / 4) registers;
//  * the tile goes from global memory STRAIGHT into LDS (global_load_lds_dwordx4: no staging registers, no LDS writes by the waves) into a ring of NBUF buffers, the
//    loads NBUF - 1 tiles ahead of the MFMAs, one raw barrier per tile behind a COUNTED vmcnt wait (only the oldest tile's loads are waited for).  A DMA writes the
//    64 lanes' 16-byte chunks to consecutive LDS addresses, so the tile image is unpadded (column = 256 B = 16 chunks) with chunk c of column j at position
//    c ^ (j & 15) -- the permutation is applied to the SOURCE address of the load and again by the operand read, whose 32 lanes of a half-wave then hit 32 different
//    bank pairs.  Columns beyond k load column k - 1 (their entries of G are never stored); the ragged last tile (n mod 32 rows) is staged by ordinary loads, zero
//    filled, by the block whose turn it is, after its loop.
// A tile of G is held in four row pieces by the waves of a group: they meet in LDS at the end, one wave after the other in a fixed order.
// Results: partial[block][slot], slot = j (k + 1) + i for i in tile row I, j in tile column J >= I -- panel_xhy_mfma's layout with
// flags = 3 (Y is X, upper tiles only), summed by finish_xhy; no norm slots.
template <int KP> struct GramRowSplit {
    static constexpr int NT = KP * (KP + 1) / 2, N0 = (NT + 1) / 2, N1 = NT - N0;
    // (an index beyond the list -- the second group of a one-tile list -- gives the last tile row instead of running away)
    static constexpr int tile_i(int idx) { int I = 0, rem = idx; while (I < KP - 1 && rem >= KP - I) { rem -= KP - I; ++I; } return I; }
    static constexpr int tile_j(int idx) { int I = 0, rem = idx; while (I < KP - 1 && rem >= KP - I) { rem -= KP - I; ++I; } return I + rem < KP ? I + rem : KP - 1; }
};
template <int KP, int FIRST, int... Q>
__device__ __forceinline__ void gram_rs_step(const double (&r)[KP], v4d (&acc)[GramRowSplit<KP>::N0], std::integer_sequence<int, Q...>) {
    using D = GramRowSplit<KP>;
    // (integral_constant: the tile coordinates must be constants BEFORE the optimiser runs, or r[] is indexed at run time -- through scratch)
    ((acc[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(r[std::integral_constant<int, D::tile_i(FIRST + Q)>::value],
                                                    r[std::integral_constant<int, D::tile_j(FIRST + Q)>::value], acc[Q], 0, 0, 0)), ...);
}
template <int KP, int FIRST, int... Q>
__device__ __forceinline__ void gram_rs_store(double *pb, int k, int arow, int acol, const v4d (&acc)[GramRowSplit<KP>::N0], std::integer_sequence<int, Q...>) {
    using D = GramRowSplit<KP>;
    auto one = [&](int I, int J, const v4d &a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + arow + 4 * r, j = 16 * J + acol;
            if (i < k && j < k) pb[(int64_t)j * (k + 1) + i] = a[r];
        }
    };
    (one(std::integral_constant<int, D::tile_i(FIRST + Q)>::value, std::integral_constant<int, D::tile_j(FIRST + Q)>::value, acc[Q]), ...);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int KP, int NBUF, int WPE>
__global__ __launch_bounds__(512, WPE) void panel_gram_rs(const double *__restrict__ X, int64_t ldx, int k, int64_t n, double *__restrict__ partial) {
    using D = GramRowSplit<KP>;
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    typedef const __attribute__((address_space(1))) void *glb_ptr_t;
    constexpr int BUFB = KP * 4096, FULL = KP / 2, ODD = KP & 1;             // bytes per tile buffer; FULL block-wide passes of 16-byte chunks, and half a pass (waves 0-3)
    extern __shared__ __attribute__((aligned(16))) double grs_lds[];           // (the ONLY LDS object of the kernel: a second one costs a vmcnt(0) in front of the operand reads)
    char *lds = reinterpret_cast<char *>(grs_lds);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int arow = lane >> 4, acol = lane & 15;
    const int grp = wave >> 2, ws = wave & 3;
    const int64_t nfull = n / 32, G = gridDim.x;
    int oa[2];                                                                 // byte offsets of this lane's operand in column block 0 on the wave's two row steps
#pragma unroll
    for (int e = 0; e < 2; ++e) oa[e] = acol * 256 + (((2 * (2 * ws + e) + (arow >> 1)) ^ acol) << 4) + (arow & 1) * 8;
    const int pcol = t >> 4, plog = (t & 15) ^ (pcol & 15);                    // the column (of a pass) and the LOGICAL chunk whose data lands at this thread's position
    v4d acc[D::N0];
#pragma unroll
    for (int q = 0; q < D::N0; ++q) acc[q] = v4d{0.0, 0.0, 0.0, 0.0};

    auto run = [&](auto first, auto seq) {
        constexpr int FIRST = decltype(first)::value;
        constexpr int LPT = FULL + ((ODD && FIRST == 0) ? 1 : 0);             // DMA instructions per tile of this wave
        auto issue = [&](int64_t Tc, int buf) {                               // tile Tc (a full one) into buffer buf
            const double *src = X + 32 * Tc + 2 * plog;
            char *dst = lds + buf * BUFB + 1024 * wave;
#pragma unroll
            for (int s = 0; s < LPT; ++s) {
                const int col = pcol + 32 * s, colc = col < k ? col : k - 1;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (int64_t)colc * ldx), (lds_ptr_t)(dst + 8192 * s), 16, 0, 2);     // (aux = 2: non-temporal, the panel is read once)
            }
        };
        // both row steps' operands first, then all the MFMAs back to back.  The reads are inline-asm ds_read_b64, ONE per operand: left to the compiler, two reads off one
        // address register become a ds_read2st64_b64, whose two elements (column blocks 4 KB apart) hit the same banks -- SQ_LDS_BANK_CONFLICT was half of the LDS cycles
        // (profiles/r06_pmc_mfma_lds.txt), zero with single reads, and the kernel 1-4 % faster.  An asm read is invisible to the compiler's wait counting: the s_waitcnt
        // below is ours, and the empty asm statements after it tie every operand to it (asm volatile statements keep their order; the MFMAs depend on the tied values).
        constexpr int B0 = std::integral_constant<int, D::tile_i(FIRST)>::value;   // the group's first column block (the blocks below it feed none of its tiles)
        auto steps = [&](const char *Xb) {
            double r[2][KP];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned a = (unsigned)(uintptr_t)(Xb + oa[e]);             // LDS byte address (the low 32 bits of the generic pointer)
#pragma unroll
                for (int b = B0; b < KP; ++b) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[e][b]) : "v"(a), "n"(4096 * b));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int b = B0; b < KP; ++b) asm volatile("" : "+v"(r[e][b]));
            gram_rs_step<KP, FIRST>(r[0], acc, seq);
            gram_rs_step<KP, FIRST>(r[1], acc, seq);
        };
        int64_t T = blockIdx.x;
        if (T < nfull) {
#pragma unroll
            for (int j = 0; j < NBUF - 1; ++j) issue(T + j * G < nfull ? T + j * G : T, j);
            int buf = 0;
            for (; T < nfull; T += G) {
                wait_vmcnt<(NBUF - 2) * LPT>();                               // this wave's loads of tile T have landed ...
                __builtin_amdgcn_s_barrier();                                 // ... everybody's have, and buffer buf - 1 has been read by all
                const int64_t Tl = T + (NBUF - 1) * G;
                issue(Tl < nfull ? Tl : T, buf == 0 ? NBUF - 1 : buf - 1);    // (beyond the panel: a tile that is never read, so that the count above holds)
                steps(lds + buf * BUFB);
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
            wait_vmcnt<0>();
        }
        __syncthreads();
        if ((n & 31) != 0 && (int64_t)blockIdx.x == nfull % G) {              // the ragged tile: ordinary loads, zero filled, into buffer 0 in the same image
            for (int p = t; p < KP * 256; p += 512) {
                const int col = p >> 4, lg = (p & 15) ^ (col & 15);
                const int64_t r0 = 32 * nfull + 2 * lg;
                v2d v = v2d{0.0, 0.0};
                if (col < k) {
                    if (r0 < n) v.x = X[(int64_t)col * ldx + r0];
                    if (r0 + 1 < n) v.y = X[(int64_t)col * ldx + r0 + 1];
                }
                *reinterpret_cast<v2d *>(lds + 16 * p) = v;
            }
            __syncthreads();
            steps(lds);
        }
    };
    // (each group runs its own copy of the whole loop: with the branch inside it the accumulators of the two paths are different values to the register
    //  allocator, which copies them back and forth and spills)
    if (grp == 0) run(std::integral_constant<int, 0>{}, std::make_integer_sequence<int, D::N0>{});
    else run(std::integral_constant<int, D::N0>{}, std::make_integer_sequence<int, D::N1>{});
    __syncthreads();
    // the four row pieces of every tile meet in the wave with ws = 0 of each group: waves 1, 2, 3, then 5, 6, 7 hand theirs over through LDS (N0 tiles of 256 doubles
    // fit the first buffer), one after the other
    double *Xt = grs_lds;
    for (int w = 1; w < 8; ++w) {
        if ((w & 3) == 0) continue;
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < D::N0; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) Xt[q * 256 + r * 64 + lane] = acc[q][r];
        }
        __syncthreads();
        if (wave == (w & 4)) {
#pragma unroll
            for (int q = 0; q < D::N0; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] += Xt[q * 256 + r * 64 + lane];
        }
        __syncthreads();
    }
    double *pb = partial + (int64_t)blockIdx.x * ((int64_t)k * (k + 1));
    if (wave == 0) gram_rs_store<KP, 0>(pb, k, arow, acol, acc, std::make_integer_sequence<int, D::N0>{});
    if (wave == 4) gram_rs_store<KP, D::N0>(pb, k, arow, acol, acc, std::make_integer_sequence<int, D::N1>{});
}

// Gram matrix G = X^H X of a COMPLEX basis of 33..80 columns: panel_gram_rs's row split and LDS-DMA tiles with panel_gram_mfma3m's three real products per complex
// one (round 6; gram_matrix, AbstractVectors.fypp:645-657).  A complex element is one 16-byte chunk, so the panel read as a REAL one of 2n rows gives the same tile image
// (32 real rows = 16 elements per column) and the same DMA code; a k-step of the MFMAs is four ELEMENTS: wave w takes element step w & 3 of the tile for half of the upper
// tile list (waves 0-3 / 4-7), reads (re, im) of each of its column blocks with ONE ds_read_b128 (inline asm, as panel_gram_rs and for its reason), forms re + im and
// im - re once per block, and runs P1 = Xr_I^T Xr_J, P2 = Xi_I^T Xi_J, P3 = (Xr + Xi)_I^T (Xi - Xr)_J for its tiles: three accumulators per tile -- 24 registers -- which
// is what limits this kernel to five column blocks (8 tiles per wave); wider complex bases keep panel_gram_mfma3m.  Re G = P1 + P2, Im G = P3 + P1 - P2, formed after the
// four element-step pieces of a tile have met in LDS.  Results: partial[block][slot], slot = (j (k + 1) + i) * 2 (+ 1), as panel_gram_mfma3m.
template <int KP, int FIRST, int... Q>
__device__ __forceinline__ void gram_rs3m_step(const double (&zr)[KP], const double (&zi)[KP], const double (&sm)[KP], const double (&df)[KP],
                                               v4d (&p1)[GramRowSplit<KP>::N0], v4d (&p2)[GramRowSplit<KP>::N0], v4d (&p3)[GramRowSplit<KP>::N0], std::integer_sequence<int, Q...>) {
    using D = GramRowSplit<KP>;
    ((p1[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zr[std::integral_constant<int, D::tile_i(FIRST + Q)>::value], zr[std::integral_constant<int, D::tile_j(FIRST + Q)>::value], p1[Q], 0, 0, 0),
      p2[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zi[std::integral_constant<int, D::tile_i(FIRST + Q)>::value], zi[std::integral_constant<int, D::tile_j(FIRST + Q)>::value], p2[Q], 0, 0, 0),
      p3[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[std::integral_constant<int, D::tile_i(FIRST + Q)>::value], df[std::integral_constant<int, D::tile_j(FIRST + Q)>::value], p3[Q], 0, 0, 0)),
     ...);
}
template <int KP, int FIRST, int... Q>
__device__ __forceinline__ void gram_rs3m_store(double *pb, int k, int arow, int acol, const v4d (&p1)[GramRowSplit<KP>::N0], const v4d (&p2)[GramRowSplit<KP>::N0],
                                                const v4d (&p3)[GramRowSplit<KP>::N0], std::integer_sequence<int, Q...>) {
    using D = GramRowSplit<KP>;
    auto one = [&](int I, int J, const v4d &a1, const v4d &a2, const v4d &a3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + arow + 4 * r, j = 16 * J + acol;
            if (i < k && j < k) {
                pb[((int64_t)j * (k + 1) + i) * 2] = a1[r] + a2[r];
                pb[((int64_t)j * (k + 1) + i) * 2 + 1] = (a3[r] + a1[r]) - a2[r];
            }
        }
    };
    (one(std::integral_constant<int, D::tile_i(FIRST + Q)>::value, std::integral_constant<int, D::tile_j(FIRST + Q)>::value, p1[Q], p2[Q], p3[Q]), ...);
}
template <int KP, int NBUF, int WPE>
__global__ __launch_bounds__(512, WPE) void panel_gram_rs3m(const double *__restrict__ X, int64_t ldx, int k, int64_t n, double *__restrict__ partial) {
    using D = GramRowSplit<KP>;
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    typedef const __attribute__((address_space(1))) void *glb_ptr_t;
    constexpr int BUFB = KP * 4096, FULL = KP / 2, ODD = KP & 1;
    extern __shared__ __attribute__((aligned(16))) double grs3_lds[];          // (the ONLY LDS object of the kernel)
    char *lds = reinterpret_cast<char *>(grs3_lds);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int arow = lane >> 4, acol = lane & 15;
    const int grp = wave >> 2, ws = wave & 3;
    const int64_t ldr = 2 * ldx, nr = 2 * n;                                   // the panel as a real one: column stride and rows
    const int64_t nfull = nr / 32, G = gridDim.x;
    const int oc = acol * 256 + (((4 * ws + arow) ^ acol) << 4);               // this lane's element (re, im) in column block 0 on the wave's element step
    const int pcol = t >> 4, plog = (t & 15) ^ (pcol & 15);
    v4d p1[D::N0], p2[D::N0], p3[D::N0];
#pragma unroll
    for (int q = 0; q < D::N0; ++q) { p1[q] = v4d{0.0, 0.0, 0.0, 0.0}; p2[q] = v4d{0.0, 0.0, 0.0, 0.0}; p3[q] = v4d{0.0, 0.0, 0.0, 0.0}; }

    auto run = [&](auto first, auto seq) {
        constexpr int FIRST = decltype(first)::value;
        constexpr int LPT = FULL + ((ODD && FIRST == 0) ? 1 : 0);
        constexpr int B0 = std::integral_constant<int, D::tile_i(FIRST)>::value;
        auto issue = [&](int64_t Tc, int buf) {
            const double *src = X + 32 * Tc + 2 * plog;
            char *dst = lds + buf * BUFB + 1024 * wave;
#pragma unroll
            for (int s = 0; s < LPT; ++s) {
                const int col = pcol + 32 * s, colc = col < k ? col : k - 1;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (int64_t)colc * ldr), (lds_ptr_t)(dst + 8192 * s), 16, 0, 2);
            }
        };
        auto steps = [&](const char *Xb) {
            v2d z[KP];
            const unsigned a = (unsigned)(uintptr_t)(Xb + oc);
#pragma unroll
            for (int b = B0; b < KP; ++b) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(z[b]) : "v"(a), "n"(4096 * b));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double zr[KP], zi[KP], sm[KP], df[KP];
#pragma unroll
            for (int b = B0; b < KP; ++b) {
                asm volatile("" : "+v"(z[b]));
                zr[b] = z[b].x; zi[b] = z[b].y; sm[b] = z[b].x + z[b].y; df[b] = z[b].y - z[b].x;
            }
#pragma unroll
            for (int b = 0; b < B0; ++b) { zr[b] = 0.0; zi[b] = 0.0; sm[b] = 0.0; df[b] = 0.0; }       // (never used)
            gram_rs3m_step<KP, FIRST>(zr, zi, sm, df, p1, p2, p3, seq);
        };
        int64_t T = blockIdx.x;
        if (T < nfull) {
#pragma unroll
            for (int j = 0; j < NBUF - 1; ++j) issue(T + j * G < nfull ? T + j * G : T, j);
            int buf = 0;
            for (; T < nfull; T += G) {
                wait_vmcnt<(NBUF - 2) * LPT>();
                __builtin_amdgcn_s_barrier();
                const int64_t Tl = T + (NBUF - 1) * G;
                issue(Tl < nfull ? Tl : T, buf == 0 ? NBUF - 1 : buf - 1);
                steps(lds + buf * BUFB);
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
            wait_vmcnt<0>();
        }
        __syncthreads();
        if ((nr & 31) != 0 && (int64_t)blockIdx.x == nfull % G) {              // the ragged tile: ordinary loads, zero filled, into buffer 0 in the same image
            for (int p = t; p < KP * 256; p += 512) {
                const int col = p >> 4, lg = (p & 15) ^ (col & 15);
                const int64_t e = 16 * nfull + lg;                             // element
                v2d v = v2d{0.0, 0.0};
                if (col < k && e < n) v = *reinterpret_cast<const v2d *>(X + (int64_t)col * ldr + 2 * e);
                *reinterpret_cast<v2d *>(lds + 16 * p) = v;
            }
            __syncthreads();
            steps(lds);
        }
    };
    if (grp == 0) run(std::integral_constant<int, 0>{}, std::make_integer_sequence<int, D::N0>{});
    else run(std::integral_constant<int, D::N0>{}, std::make_integer_sequence<int, D::N1>{});
    __syncthreads();
    // the four element-step pieces of every tile meet in the wave with ws = 0 of each group, one wave after the other (3 N0 tiles of 256 doubles fit the ring)
    double *Xt = grs3_lds;
    for (int w = 1; w < 8; ++w) {
        if ((w & 3) == 0) continue;
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < D::N0; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Xt[(3 * q + 0) * 256 + r * 64 + lane] = p1[q][r];
                    Xt[(3 * q + 1) * 256 + r * 64 + lane] = p2[q][r];
                    Xt[(3 * q + 2) * 256 + r * 64 + lane] = p3[q][r];
                }
        }
        __syncthreads();
        if (wave == (w & 4)) {
#pragma unroll
            for (int q = 0; q < D::N0; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p1[q][r] += Xt[(3 * q + 0) * 256 + r * 64 + lane];
                    p2[q][r] += Xt[(3 * q + 1) * 256 + r * 64 + lane];
                    p3[q][r] += Xt[(3 * q + 2) * 256 + r * 64 + lane];
                }
        }
        __syncthreads();
    }
    double *pb = partial + (int64_t)blockIdx.x * ((int64_t)k * (k + 1) * 2);
    if (wave == 0) gram_rs3m_store<KP, 0>(pb, k, arow, acol, p1, p2, p3, std::make_integer_sequence<int, D::N0>{});
    if (wave == 4) gram_rs3m_store<KP, D::N0>(pb, k, arow, acol, p1, p2, p3, std::make_integer_sequence<int, D::N1>{});
}

// panel_gram_rs3m for 81..112 complex columns (six or seven column blocks): FOUR groups of two waves, each a quarter of the upper-tile list (5-7 tiles: 120-168
// accumulator registers) on two of the tile's four element steps per wave.  Group g = wave >> 1: SIMD s runs the waves s and s + 4, i.e. the groups (0, 2) or (1, 3) --
// the list is cut so that |g0| + |g2| and |g1| + |g3| differ by at most one tile.  Everything else as panel_gram_rs3m.
template <int KP> struct GramRowSplit4 {
    static constexpr int NT = KP * (KP + 1) / 2, BASE = NT / 4, REM = NT % 4, NM = BASE + (REM ? 1 : 0);
    static constexpr int first(int g) { return g * BASE + (g < REM ? g : REM); }
    static constexpr int count(int g) { return BASE + (g < REM ? 1 : 0); }
};
template <int KP, int NM, int FIRST, int... Q>
__device__ __forceinline__ void gram3m_step(const double (&zr)[KP], const double (&zi)[KP], const double (&sm)[KP], const double (&df)[KP], v4d (&p1)[NM], v4d (&p2)[NM],
                                            v4d (&p3)[NM], std::integer_sequence<int, Q...>) {
    using D = GramRowSplit<KP>;
    ((p1[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zr[std::integral_constant<int, D::tile_i(FIRST + Q)>::value], zr[std::integral_constant<int, D::tile_j(FIRST + Q)>::value], p1[Q], 0, 0, 0),
      p2[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(zi[std::integral_constant<int, D::tile_i(FIRST + Q)>::value], zi[std::integral_constant<int, D::tile_j(FIRST + Q)>::value], p2[Q], 0, 0, 0),
      p3[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(sm[std::integral_constant<int, D::tile_i(FIRST + Q)>::value], df[std::integral_constant<int, D::tile_j(FIRST + Q)>::value], p3[Q], 0, 0, 0)),
     ...);
}
// (tile by tile: the partner wave's pieces are added and the tile stored before the next one is touched -- all adds first, then all stores, spilled in the epilogue)
template <int KP, int NM, int FIRST, int... Q>
__device__ __forceinline__ void gram3m_add_store(double *pb, int k, int arow, int acol, int lane, const double *Xt, const v4d (&p1)[NM], const v4d (&p2)[NM], const v4d (&p3)[NM],
                                                 std::integer_sequence<int, Q...>) {
    using D = GramRowSplit<KP>;
    auto one = [&](int q, int I, int J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double a1 = p1[q][r] + Xt[(3 * q + 0) * 256 + r * 64 + lane], a2 = p2[q][r] + Xt[(3 * q + 1) * 256 + r * 64 + lane],
                         a3 = p3[q][r] + Xt[(3 * q + 2) * 256 + r * 64 + lane];
            const int i = 16 * I + arow + 4 * r, j = 16 * J + acol;
            if (i < k && j < k) {
                pb[((int64_t)j * (k + 1) + i) * 2] = a1 + a2;
                pb[((int64_t)j * (k + 1) + i) * 2 + 1] = (a3 + a1) - a2;
            }
        }
    };
    (one(Q, std::integral_constant<int, D::tile_i(FIRST + Q)>::value, std::integral_constant<int, D::tile_j(FIRST + Q)>::value), ...);
}
template <int KP, int NBUF, int WPE>
__global__ __launch_bounds__(512, WPE) void panel_gram_rs3m4(const double *__restrict__ X, int64_t ldx, int k, int64_t n, double *__restrict__ partial) {
    using D = GramRowSplit<KP>;
    using D4 = GramRowSplit4<KP>;
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    typedef const __attribute__((address_space(1))) void *glb_ptr_t;
    constexpr int BUFB = KP * 4096, FULL = KP / 2, ODD = KP & 1, NM = D4::NM;
    extern __shared__ __attribute__((aligned(16))) double grs34_lds[];         // (the ONLY LDS object of the kernel)
    char *lds = reinterpret_cast<char *>(grs34_lds);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int arow = lane >> 4, acol = lane & 15;
    const int grp = wave >> 1, half = wave & 1;
    const int64_t ldr = 2 * ldx, nr = 2 * n;                                   // the panel as a real one: column stride and rows
    const int64_t nfull = nr / 32, G = gridDim.x;
    int oc[2];                                                                 // this lane's element (re, im) in column block 0 on the wave's two element steps
#pragma unroll
    for (int e = 0; e < 2; ++e) oc[e] = acol * 256 + (((4 * (2 * half + e) + arow) ^ acol) << 4);
    const int pcol = t >> 4, plog = (t & 15) ^ (pcol & 15);
    v4d p1[NM], p2[NM], p3[NM];
#pragma unroll
    for (int q = 0; q < NM; ++q) { p1[q] = v4d{0.0, 0.0, 0.0, 0.0}; p2[q] = v4d{0.0, 0.0, 0.0, 0.0}; p3[q] = v4d{0.0, 0.0, 0.0, 0.0}; }

    auto run = [&](auto gc) {
        constexpr int GI = decltype(gc)::value, FIRST = D4::first(GI), CNT = D4::count(GI);
        constexpr int LPT = FULL + ((ODD && GI < 2) ? 1 : 0);                  // (the half pass of an odd number of column blocks is issued by the waves 0-3 = groups 0, 1)
        constexpr int B0 = std::integral_constant<int, D::tile_i(FIRST)>::value;
        constexpr auto seq = std::make_integer_sequence<int, CNT>{};
        auto issue = [&](int64_t Tc, int buf) {
            const double *src = X + 32 * Tc + 2 * plog;
            char *dst = lds + buf * BUFB + 1024 * wave;
#pragma unroll
            for (int s = 0; s < LPT; ++s) {
                const int col = pcol + 32 * s, colc = col < k ? col : k - 1;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (int64_t)colc * ldr), (lds_ptr_t)(dst + 8192 * s), 16, 0, 2);
            }
        };
        auto steps = [&](const char *Xb) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v2d z[KP];
                const unsigned a = (unsigned)(uintptr_t)(Xb + oc[e]);
#pragma unroll
                for (int b = B0; b < KP; ++b) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(z[b]) : "v"(a), "n"(4096 * b));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double zr[KP], zi[KP], sm[KP], df[KP];
#pragma unroll
                for (int b = B0; b < KP; ++b) {
                    asm volatile("" : "+v"(z[b]));
                    zr[b] = z[b].x; zi[b] = z[b].y; sm[b] = z[b].x + z[b].y; df[b] = z[b].y - z[b].x;
                }
#pragma unroll
                for (int b = 0; b < B0; ++b) { zr[b] = 0.0; zi[b] = 0.0; sm[b] = 0.0; df[b] = 0.0; }   // (never used)
                gram3m_step<KP, NM, FIRST>(zr, zi, sm, df, p1, p2, p3, seq);
                __builtin_amdgcn_sched_barrier(0);                             // (or the second step's reads move ahead of these MFMAs and both operand sets are live: spills)
            }
        };
        int64_t T = blockIdx.x;
        if (T < nfull) {
#pragma unroll
            for (int j = 0; j < NBUF - 1; ++j) issue(T + j * G < nfull ? T + j * G : T, j);
            int buf = 0;
            for (; T < nfull; T += G) {
                wait_vmcnt<(NBUF - 2) * LPT>();
                __builtin_amdgcn_s_barrier();
                const int64_t Tl = T + (NBUF - 1) * G;
                issue(Tl < nfull ? Tl : T, buf == 0 ? NBUF - 1 : buf - 1);
                steps(lds + buf * BUFB);
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
            wait_vmcnt<0>();
        }
        __syncthreads();
        if ((nr & 31) != 0 && (int64_t)blockIdx.x == nfull % G) {              // the ragged tile: ordinary loads, zero filled, into buffer 0 in the same image
            for (int p = t; p < KP * 256; p += 512) {
                const int col = p >> 4, lg = (p & 15) ^ (col & 15);
                const int64_t e = 16 * nfull + lg;
                v2d v = v2d{0.0, 0.0};
                if (col < k && e < n) v = *reinterpret_cast<const v2d *>(X + (int64_t)col * ldr + 2 * e);
                *reinterpret_cast<v2d *>(lds + 16 * p) = v;
            }
            __syncthreads();
            steps(lds);
        }
        __syncthreads();
        // the two halves of every tile meet in the group's even wave, through LDS: 3 NM tiles of 256 doubles per group -- two groups at a time fit the ring
        double *Xt = grs34_lds + (GI & 1) * 3 * NM * 256;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if ((GI >> 1) == rnd && half == 1) {
#pragma unroll
                for (int q = 0; q < CNT; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Xt[(3 * q + 0) * 256 + r * 64 + lane] = p1[q][r];
                        Xt[(3 * q + 1) * 256 + r * 64 + lane] = p2[q][r];
                        Xt[(3 * q + 2) * 256 + r * 64 + lane] = p3[q][r];
                    }
            }
            __syncthreads();
            if ((GI >> 1) == rnd && half == 0)
                gram3m_add_store<KP, NM, FIRST>(partial + (int64_t)blockIdx.x * ((int64_t)k * (k + 1) * 2), k, arow, acol, lane, Xt, p1, p2, p3, seq);
            __syncthreads();
        }
    };
    // (each group runs its own copy of the whole loop: see panel_gram_rs)
    if (grp == 0) run(std::integral_constant<int, 0>{});
    else if (grp == 1) run(std::integral_constant<int, 1>{});
    else if (grp == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
}

// Pass B of the block Gram-Schmidt with many right-hand sides, FUSED on the matrix cores (gram_schmidt.fypp:59-105):
//     Y' = Y - X H1   (stored)      M2 = X^H Y'      ||Y'_q||^2
// in ONE pass over X(:, :k) (k <= 128) and Y(:, :p) (p <= 32) -- the update and the second coefficient pass of
// DGS_basis_against_basis used to be two passes (MFMA update, then panel_xhy_mfma), four per group of 32 columns in all;
// with this kernel a group costs three (panel_xhy_mfma | this | MFMA update).
// A block of FOUR waves stages a tile of 32 real rows x all columns of X and Y in LDS as panel_xhy_mfma does (same strides, the
// next tile prefetched into registers), plus -- once -- the coefficients H1 as B operands: 78 KB for k = 128, p = 32 in the
// real kind, so two blocks share a CU and one block's staging / barriers / stores run under the other's MFMAs (a first version
// with one 8-wave block per CU and 64-row tiles spent 40 % of its time outside the matrix pipe).  Per tile:
//   update : D(16 rows x 16 rhs) += A(16 rows x 4 cols) B(4 cols x 16 rhs), wave w owning row block w & 1 and rhs tile w >> 1 of
//            the 32 x 32 tile of U = X H1.  A k-step takes the columns c = 32 g + t + 8 kk (kk = 0..3), not 4 consecutive ones:
//            with the column stride S = 34 (== 2 mod 32) that makes the A read (16 rows x 4 columns) and the B read of H1
//            conflict free for the two 32-lane groups of a ds_read_b64.  Y' = Y - U goes back INTO the LDS tile;
//   store  : every thread writes the 16 bytes of Y' it staged (coalesced along the rows) and adds them to the norms;
//   dots   : as panel_xhy_mfma (wave w owns tile rows I = w and w + 4 of X, contraction over the 32 rows).
// complex kind: the panel is read as a real one of 2n rows (re, im interleaved).  U(rho, q) = sum_c X(rho, c) Hr(c, q) + s(rho)
// X(rho^1, c) Hi(c, q), s = -1 on the even (real-part) rows: a second MFMA per k-step whose A operand is the partner row
// with that sign and whose B operand is the imaginary plane of H1 (113 KB: one block per CU -- the launcher keeps the
// four-pass schedule for the complex kind, which is faster there).
// H1: device coefficients in panel_dot_p's layout [q][k + 1][ED].  Results as panel_xhy_mfma (partial[block][slot], norms in
// npartial[block][q]).
template <bool CPLX>
__global__ __launch_bounds__(256, CPLX ? 1 : 2) void panel_xhy_upd_mfma(const double *__restrict__ X, int64_t ldx, int k,
                                                          double *__restrict__ Y, int64_t ldy, int p, int64_t n,
                                                          const double *__restrict__ H1, double *__restrict__ partial,
                                                          double *__restrict__ npartial, int policy, Guard guard) {
    if (stopped(guard)) return;
    constexpr int ER = CPLX ? 2 : 1;
    constexpr int NT = 256;                                                   // threads per block
    constexpr int TR = 32, S = TR + 2, CH = TR / 2, CHS = 4, CPP = NT / CH;   // 16 columns staged per block-wide pass
    constexpr int PJM = 2, NXP = 128 / CPP, NYP = (16 * PJM + CPP - 1) / CPP;
    constexpr int RL = 16 * PJM + 2;             // row length of the staged coefficients (8 RL == 16 mod 32)
    extern __shared__ __attribute__((aligned(16))) double xu_lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int KP = (k + 15) >> 4, PJ = (p + 15) >> 4, KG = (k + 31) >> 5;
    const int KS = (k + CPP - 1) / CPP, PS = (p + CPP - 1) / CPP;
    double *Xt = xu_lds;                                     // [32 KG columns][S]  (columns >= 16 KP stay zero)
    double *Yt = Xt + KG * 32 * S;                           // [16 PJM columns][S]
    double *Hr = Yt + 16 * PJM * S;                          // [32 KG columns][RL]  (+ the imaginary plane for the complex kind)
    double *Hi = Hr + KG * 32 * RL;
    const int64_t nr = n * ER;
    const int64_t ntiles = (nr + TR - 1) / TR;
    const int64_t xcs = ldx * ER, ycs = ldy * ER;
    const int arow = lane >> 4, acol = lane & 15;

    // ---- once: zero the padding columns of X's tile, stage H1 (zero beyond k / p)
    for (int i = t; i < (KG * 32 - KP * 16) * S; i += NT) Xt[KP * 16 * S + i] = 0.0;
    for (int i = t; i < KG * 32 * RL; i += NT) {
        const int c = i / RL, q = i - c * RL;
        double hr = 0.0, hi = 0.0;
        if (c < k && q < p) {
            hr = H1[((int64_t)q * (k + 1) + c) * ER];
            if constexpr (CPLX) hi = H1[((int64_t)q * (k + 1) + c) * ER + 1];
        }
        Hr[i] = hr;
        if constexpr (CPLX) Hi[i] = hi;
    }

    v4d acc_re[2][PJM], acc_im[2][PJM];          // tile rows I = wave, wave + 4 of X
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int J = 0; J < PJM; ++J) { acc_re[h][J] = v4d{0.0, 0.0, 0.0, 0.0}; acc_im[h][J] = v4d{0.0, 0.0, 0.0, 0.0}; }
    double nacc[NYP];
#pragma unroll
    for (int s = 0; s < NYP; ++s) nacc[s] = 0.0;
    v2d xs[NXP], ys[NYP];

#ifdef LK_DIAGNOSTICS
    const int dbg = policy >> 4;                    // diagnostics build only (upd_debug): WRONG results, phase timing
#else
    constexpr int dbg = 0;
#endif
    policy &= 15;
    auto gload = [&](int64_t T) {
        if ((dbg & 4) && T != (int64_t)blockIdx.x) return;
        const int64_t rbase = T * TR;
#pragma unroll
        for (int s = 0; s < NXP; ++s) {
            xs[s] = v2d{0.0, 0.0};
            if (s < KS) {
                const int c = t + NT * s, col = c >> CHS;
                const int64_t rr = rbase + 2 * (c & (CH - 1));
                if (col < k) {
                    const double *pc = X + (int64_t)col * xcs;
                    if (rr + 1 < nr) xs[s] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(pc + rr));
                    else if (rr < nr) xs[s].x = pc[rr];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NYP; ++s) {
            ys[s] = v2d{0.0, 0.0};
            if (s < PS) {
                const int c = t + NT * s, col = c >> CHS;
                const int64_t rr = rbase + 2 * (c & (CH - 1));
                if (col < p) {
                    const double *pc = Y + (int64_t)col * ycs;
                    if (rr + 1 < nr) ys[s] = *reinterpret_cast<const v2d *>(pc + rr);
                    else if (rr < nr) ys[s].x = pc[rr];
                }
            }
        }
    };

    int64_t T = blockIdx.x;
    if (T < ntiles) gload(T);
    for (; T < ntiles; T += gridDim.x) {
        __syncthreads();                                            // the previous tile's operands have been read
#pragma unroll
        for (int s = 0; s < NXP; ++s)
            if (s < KS) {
                const int c = t + NT * s;
                if ((c >> CHS) < KP * 16) *reinterpret_cast<v2d *>(Xt + (c >> CHS) * S + 2 * (c & (CH - 1))) = xs[s];
            }
#pragma unroll
        for (int s = 0; s < NYP; ++s)
            if (s < PS) {
                const int c = t + NT * s;
                if ((c >> CHS) < PJ * 16) *reinterpret_cast<v2d *>(Yt + (c >> CHS) * S + 2 * (c & (CH - 1))) = ys[s];
            }
        __syncthreads();
        const int64_t Tcur = T;
        if (T + gridDim.x < ntiles) gload(T + gridDim.x);           // in flight while this tile's MFMAs run

        // ---- update: this wave's 16 x 16 tile of U = X H1, then Y' = Y - U into the LDS tile
        {
            const int rb = wave & 1, J = wave >> 1;
            if (J < PJ && !(dbg & 1)) {
                // four independent accumulator chains (k-steps tt, tt + 4 of every column group share one)
                v4d ua[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ua[i] = v4d{0.0, 0.0, 0.0, 0.0};
                const int r0 = rb * 16 + acol;                     // A operand: row of the tile (lane & 15), column step kk = lane >> 4
                for (int g = 0; g < KG; ++g) {
                    double av[8], bv[8], av2[CPLX ? 8 : 1], bv2[CPLX ? 8 : 1];
#pragma unroll
                    for (int tt = 0; tt < 8; ++tt) {               // all LDS reads of the column group first, then its MFMAs
                        const int c = g * 32 + tt + 8 * arow;
                        av[tt] = Xt[c * S + r0];
                        bv[tt] = Hr[c * RL + J * 16 + acol];
                        if constexpr (CPLX) {
                            const double a2 = Xt[c * S + (r0 ^ 1)];
                            av2[tt] = (r0 & 1) ? a2 : -a2;
                            bv2[tt] = Hi[c * RL + J * 16 + acol];
                        }
                    }
#pragma unroll
                    for (int tt = 0; tt < 8; ++tt) {
                        ua[tt & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tt], bv[tt], ua[tt & 3], 0, 0, 0);
                        if constexpr (CPLX) ua[tt & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av2[tt], bv2[tt], ua[tt & 3], 0, 0, 0);
                    }
                }
                const v4d u = (ua[0] + ua[1]) + (ua[2] + ua[3]);
                // D[row = (lane >> 4) + 4 reg][col = lane & 15]: rows of the tile, right-hand side J*16 + acol
                double *yc = Yt + (J * 16 + acol) * S + rb * 16 + arow;
#pragma unroll
                for (int r = 0; r < 4; ++r) yc[4 * r] -= u[r];
            }
        }
        __syncthreads();                                            // Y' is complete in LDS

        // ---- store Y' (every thread the 16 bytes it staged) and its norms
        {
            const int64_t rbase = Tcur * TR;
#pragma unroll
            for (int s = 0; s < NYP; ++s)
                if (s < PS) {
                    const int c = t + NT * s, col = c >> CHS;
                    if (col < p) {
                        const v2d v = *reinterpret_cast<const v2d *>(Yt + col * S + 2 * (c & (CH - 1)));
                        const int64_t rr = rbase + 2 * (c & (CH - 1));
                        double *pc = Y + (int64_t)col * ycs;
                        if (dbg & 8) { nacc[s] += v.x * v.x + v.y * v.y; }
                        else if (rr + 1 < nr) { store16(reinterpret_cast<v2d *>(pc + rr), v, policy); nacc[s] += v.x * v.x + v.y * v.y; }
                        else if (rr < nr) { pc[rr] = v.x; nacc[s] += v.x * v.x; }
                    }
                }
        }

        // ---- dots: M2 += X_tile^H Y'_tile (tile rows I = wave, wave + 4 of X); the LDS reads of four row steps, then their MFMAs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int I = wave + 4 * h;
            if (I < KP && !(dbg & 2)) {
                for (int s0 = 0; s0 < TR / 4; s0 += 4) {
                    double a[4], b[4][PJM], b2[4][CPLX ? PJM : 1];
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) {
                        const int ro = 4 * (s0 + ss) + arow;
                        a[ss] = Xt[(16 * I + acol) * S + ro];
#pragma unroll
                        for (int J = 0; J < PJM; ++J) {
                            if (J < PJ) {
                                b[ss][J] = Yt[(16 * J + acol) * S + ro];
                                if constexpr (CPLX) {
                                    const double v = Yt[(16 * J + acol) * S + (ro ^ 1)];
                                    b2[ss][J] = (ro & 1) ? -v : v;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) {
#pragma unroll
                        for (int J = 0; J < PJM; ++J) {
                            if (J < PJ) {
                                acc_re[h][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ss], b[ss][J], acc_re[h][J], 0, 0, 0);
                                if constexpr (CPLX) acc_im[h][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ss], b2[ss][J], acc_im[h][J], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }

    const int64_t nslots = (int64_t)p * (k + 1) * ER;
    double *pb = partial + (int64_t)blockIdx.x * nslots;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int I = wave + 4 * h;
        if (I < KP) {
#pragma unroll
            for (int J = 0; J < PJM; ++J) {
                if (J < PJ) {
                    const int q = 16 * J + acol;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * I + arow + 4 * r;
                        if (i < k && q < p) {
                            pb[((int64_t)q * (k + 1) + i) * ER] = acc_re[h][J][r];
                            if constexpr (CPLX) pb[((int64_t)q * (k + 1) + i) * ER + 1] = acc_im[h][J][r];
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NYP; ++s) {
        if (s < PS) {
            double v = nacc[s];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            const int col = (t >> CHS) + CPP * s;
            if ((t & (CH - 1)) == 0 && col < p) npartial[(int64_t)blockIdx.x * p + col] = v;
        }
    }
}

// Pass B of the block Gram-Schmidt, real kind, 17..32 right-hand sides -- ROW-OWNER waves on LDS-DMA tiles (round 6; gram_schmidt.fypp:59-105):
//     Y' = Y - X H1   (stored)      M2 = X^T Y'      ||Y'_q||^2
// panel_xhy_upd_mfma<false> above spends its time in the LDS-fed MFMA phase (coefficients read from LDS for every update MFMA, Y' written back into the LDS tile and read
// again as the dot product's operand, three barriers per tile, register staging).  Here a TEAM of four waves takes a 32-row tile and wave w of it OWNS the 16 rows
// rb = w & 1 of the 16 right-hand sides cb = w >> 1:
//   update : u = Y_tile - X_tile H1 for those 16 x 16 entries: 4 KP MFMAs in ONE accumulator (started from the staged Y) whose B operands -- the wave's 16 columns of -H1 --
//            sit in REGISTERS for the whole kernel (8 KP of them), the A operand X(row, column) read straight from the tile;
//   store  : Y' leaves from the accumulator registers (row = (lane >> 4) + 4 r: the four stores of a lane quartet fill a 128-byte line of one column in L2);
//   dots   : the accumulator layout of v_mfma_f64_16x16x4 (register r = rows (lane >> 4) + 4 r) IS the B-operand layout of its k-step r, so M2(I, cb) += X_I^T Y' takes
//            the Y' registers as they are -- no LDS round trip, no barrier: 4 KP MFMAs on KP accumulators, A operands from the tile.
// A block is TWO teams (eight waves, two per SIMD: with one, the wave's DMA issue, LDS latencies and stores leave the matrix pipe idle -- 4.1 ms against 3.5 at k = 128,
// p = 32, n = 10^7), each on its own tile of the block's sequence and its own two stages of LDS (X: KP x 4 KB, Y: 8 KB; panel_gram_rs's image: unpadded columns, chunk c of
// column j at position c ^ (j & 15), the permutation on the source address and on every operand read): 160 KB at k = 128, ONE barrier per iteration for both teams, the
// team's next tile fetched by LDS-DMA a whole iteration ahead (dealing its ten DMA instructions per wave out between the MFMAs measured the same at k = 128 and
// slower for narrow bases: not kept).
// (The Y' stores share the vmcnt counter with the DMA loads: the wait at the head of an iteration is vmcnt(0), both are an iteration or half of one old by then.)
// A k-step of the update takes the columns 16 g + t + 8 (kk & 1) + 4 (kk >> 1), kk = lane >> 4: the two columns of a half-wave differ in bit 3, so their 16-row reads fall
// into different halves of the bank row.  Rows beyond n: the ragged last tile is staged by ordinary loads, zero filled, after the loop.
// Phases at k = 128, p = 32, n = 10^7 (knock-out builds, docs/TUNING_LOG.md): MFMAs alone 2.9 ms (0.73 of the FP64 matrix peak), loads + stores alone 2.9 ms (5.2 TB/s: the
// bytes in flight are bounded by the LDS ring -- half of 160 KB per CU), together 3.4; accumulator chains (1 / 2 / 4) make no difference.
// H1: device coefficients in panel_dot_p's layout [q][k + 1].  Results: partial[2 block + rb][slot], npartial[(2 block + rb) p + q] (finish_xhy with 2 grid partial blocks).
template <int KP>
__global__ __launch_bounds__(512, 2) void panel_xhy_upd_rs(const double *__restrict__ X, int64_t ldx, int k, double *__restrict__ Y, int64_t ldy, int p, int64_t n,
                                                           const double *__restrict__ H1, double *__restrict__ partial, double *__restrict__ npartial, Guard guard) {
    if (stopped(guard)) return;
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    typedef const __attribute__((address_space(1))) void *glb_ptr_t;
    constexpr int XB = KP * 4096, STAGE = XB + 8192;                               // bytes of X's tile / of a stage (X's tile, then Y's)
    extern __shared__ __attribute__((aligned(16))) double xur_lds[];               // (the ONLY LDS object of the kernel): stages [team][2]
    char *lds = reinterpret_cast<char *>(xur_lds);
    const int t = threadIdx.x, lane = t & 63, tt4 = t & 255;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int arow = lane >> 4, acol = lane & 15;
    const int team = wave >> 2, wv = wave & 3, rb = wv & 1, cb = wv >> 1;
    const int64_t nfull = n / 32, G = gridDim.x;
    // operand offsets inside a stage: od[r] -- rows 16 rb + arow + 4 r of column (16 I + acol) (+ 4096 I): the dot products' A operands and the staged Y in accumulator
    // layout; ou[t] -- row 16 rb + acol of column 16 g + m, m = t + 8 (arow & 1) + 4 (arow >> 1) (+ 4096 g): the update's A operands
    int od[4], ou[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) od[r] = acol * 256 + (((8 * rb + 2 * r + (arow >> 1)) ^ acol) << 4) + (arow & 1) * 8;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int m = tt + 8 * (arow & 1) + 4 * (arow >> 1);
        ou[tt] = m * 256 + (((8 * rb + (acol >> 1)) ^ m) << 4) + (acol & 1) * 8;
    }
    // the wave's columns of -H1 as B operands (zero beyond k / p)
    double hb[KP][4];
    const int q = 16 * cb + acol;
#pragma unroll
    for (int g = 0; g < KP; ++g)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int c = 16 * g + tt + 8 * (arow & 1) + 4 * (arow >> 1);
            hb[g][tt] = (c < k && q < p) ? -H1[(int64_t)q * (k + 1) + c] : 0.0;
        }
    v4d m2[KP];
#pragma unroll
    for (int I = 0; I < KP; ++I) m2[I] = v4d{0.0, 0.0, 0.0, 0.0};
    double nacc = 0.0;
    const int pcol = tt4 >> 4, plog = (tt4 & 15) ^ (pcol & 15);                   // the column (of a pass) and the LOGICAL chunk whose data lands at this thread's position
    char *const st0 = lds + team * 2 * STAGE;                                     // the team's two stages
    auto issue = [&](int64_t Tc, int buf) {                                       // tile Tc (a full one) into the team's stage buf
        char *dst = st0 + buf * STAGE + 1024 * wv;
        const double *sx = X + 32 * Tc + 2 * plog;
#pragma unroll
        for (int s = 0; s < KP; ++s) {
            const int col = pcol + 16 * s, colc = col < k ? col : k - 1;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(sx + (int64_t)colc * ldx), (lds_ptr_t)(dst + 4096 * s), 16, 0, 2);
        }
        const double *sy = Y + 32 * Tc + 2 * plog;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int col = pcol + 16 * s, colc = col < p ? col : p - 1;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(sy + (int64_t)colc * ldy), (lds_ptr_t)(dst + XB + 4096 * s), 16, 0, 0);
        }
    };
    // (Operand reads left to the compiler.  It merges pairs of them into ds_read2st64_b64, whose two elements -- column blocks 4 KB apart -- hit the same banks: a third
    //  of this kernel's LDS cycles are bank conflicts (SQ_LDS_BANK_CONFLICT, profiles/r06_pmc_mfma_lds.txt).  Single inline-asm reads pipelined with counted lgkmcnt
    //  waits were built: not faster, and WRONG at some shapes -- the compiler's scalar loads share that counter and return out of order, so a count is no guarantee.)
    auto update = [&](const char *st) -> v4d {                                    // Y' (this wave's 16 x 16 entries) of the tile in stage st
        v4d u;
        const char *yt = st + XB + 4096 * cb;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = *reinterpret_cast<const double *>(yt + od[r]);
#pragma unroll
        for (int g = 0; g < KP; ++g) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                u = __builtin_amdgcn_mfma_f64_16x16x4f64(*reinterpret_cast<const double *>(st + ou[tt] + 4096 * g), hb[g][tt], u, 0, 0, 0);
        }
        return u;
    };
    auto dots = [&](const char *st, const v4d &yp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int I = 0; I < KP; ++I)
                m2[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(*reinterpret_cast<const double *>(st + od[r] + 4096 * I), yp[r], m2[I], 0, 0, 0);
        }
    };
    double *yq = Y + (int64_t)(q < p ? q : p - 1) * ldy + 16 * rb + arow;         // this lane's column of Y, at its first row of a tile

    // the block's tiles T_j = blockIdx + j G; team tm takes j = tm, tm + 2, ...: one tile per team and iteration, ONE barrier per iteration for both teams
    const int64_t nmine = (int64_t)blockIdx.x < nfull ? (nfull - 1 - blockIdx.x) / G + 1 : 0;       // full tiles of this block
    const int64_t niter = (nmine + 1) / 2;
    if (team < nmine) issue(blockIdx.x + (int64_t)team * G, 0);
    for (int64_t it = 0; it < niter; ++it) {
        const int64_t j = 2 * it + team;
        const int buf = (int)(it & 1);
        wait_vmcnt<0>();                                                          // this wave's loads of tile j (and the stores of tile j - 2) are done ...
        __builtin_amdgcn_s_barrier();                                             // ... everybody's are, and the team's other stage has been read by all
        if (j + 2 < nmine) issue(blockIdx.x + (j + 2) * G, buf ^ 1);             // the team's next tile: a whole iteration ahead of its use
        if (j < nmine) {
            const int64_t T = blockIdx.x + j * G;
            const char *st = st0 + buf * STAGE;
            const v4d yp = update(st);
            if (q < p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) yq[32 * T + 4 * r] = yp[r];
                nacc += (yp[0] * yp[0] + yp[1] * yp[1]) + (yp[2] * yp[2] + yp[3] * yp[3]);
            }
            dots(st, yp);
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    if ((n & 31) != 0 && (int64_t)blockIdx.x == nfull % G) {                      // the ragged tile: ordinary loads, zero filled, into team 0's stage 0 in the same image
        for (int c = t; c < (KP + 2) * 256; c += 512) {
            const bool isx = c < KP * 256;
            const int cc = isx ? c : c - KP * 256;
            const int col = cc >> 4, lg = (cc & 15) ^ (col & 15);
            const int64_t r0 = 32 * nfull + 2 * lg;
            const double *src = isx ? X + (int64_t)col * ldx : Y + (int64_t)col * ldy;
            v2d v = v2d{0.0, 0.0};
            if (col < (isx ? k : p)) {
                if (r0 < n) v.x = src[r0];
                if (r0 + 1 < n) v.y = src[r0 + 1];
            }
            *reinterpret_cast<v2d *>(lds + (isx ? 0 : XB) + 16 * cc) = v;
        }
        __syncthreads();
        if (team == 0) {
            const v4d yp = update(lds);
            if (q < p) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (32 * nfull + 16 * rb + arow + 4 * r < n) yq[32 * nfull + 4 * r] = yp[r];
                nacc += (yp[0] * yp[0] + yp[1] * yp[1]) + (yp[2] * yp[2] + yp[3] * yp[3]);  // (the rows beyond n are zero: X and Y were zero filled)
            }
            dots(lds, yp);
        }
        __syncthreads();
    }
    // team 1 hands its sums to team 0 through LDS (wave wv's: KP tiles of 256 doubles and the lane's norm sum), then team 0 writes the block's two partial blocks
    if (team == 1) {
#pragma unroll
        for (int I = 0; I < KP; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r) xur_lds[(wv * (KP + 1) + I) * 256 + r * 64 + lane] = m2[I][r];
        xur_lds[(wv * (KP + 1) + KP) * 256 + lane] = nacc;
    }
    __syncthreads();
    if (team == 1) return;
#pragma unroll
    for (int I = 0; I < KP; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) m2[I][r] += xur_lds[(wv * (KP + 1) + I) * 256 + r * 64 + lane];
    nacc += xur_lds[(wv * (KP + 1) + KP) * 256 + lane];
    const int64_t nslots = (int64_t)p * (k + 1);
    double *pb = partial + (2 * (int64_t)blockIdx.x + rb) * nslots;               // (wave (rb, cb) writes the slots of its right-hand sides 16 cb ..: all of them, between the two)
#pragma unroll
    for (int I = 0; I < KP; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + arow + 4 * r;
            if (i < k && q < p) pb[(int64_t)q * (k + 1) + i] = m2[I][r];
        }
    double v = nacc;                                                               // the four lanes of a column (arow = 0..3) hold its rows
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (arow == 0 && q < p) npartial[(2 * (int64_t)blockIdx.x + rb) * p + q] = v;
}

// out[slot] = sum over vb of partial[vb][slot]; norm slots (i = k) from npartial; tiles panel_xhy_mfma skipped (flag 2:
// I > J) read as zero.  16 slots x 16 lanes per block: lane v adds the entries vb = v, v + 16, ... in order, then the 16 lane
// sums are added in lane order -- a fixed summation order whatever the grid (there can be thousands of partial blocks for a
// handful of slots: one thread per slot would walk them as one serial chain).
__global__ __launch_bounds__(256) void finish_xhy(const double *__restrict__ partial, int nvb, const double *__restrict__ npartial,
                                                  int nblocks, int k, int p, int ER, int flags, double *__restrict__ out) {
    __shared__ double sums[16][17];
    const int64_t nslots = (int64_t)p * (k + 1) * ER;
    const int sl = threadIdx.x & 15, v = threadIdx.x >> 4;
    const int64_t idx = (int64_t)blockIdx.x * 16 + sl;
    double s = 0.0;
    if (idx < nslots) {
        const int part = (int)(idx % ER), i = (int)((idx / ER) % (k + 1)), q = (int)(idx / ((int64_t)ER * (k + 1)));
        if (i == k) {
            if (part == 0 && !(flags & 1))
                for (int b = v; b < nblocks; b += 16) s += npartial[(int64_t)b * p + q];
        } else if (!((flags & 2) && (i >> 4) > (q >> 4))) {
            for (int vb = v; vb < nvb; vb += 16) s += partial[(int64_t)vb * nslots + idx];
        }
    }
    sums[v][sl] = s;
    __syncthreads();
    if (v == 0 && idx < nslots) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += sums[w][sl];
        out[idx] = tot;
    }
}

// per-lane coefficient tiles of panel_gemm_mfma from device coefficients laid out [q][ldc][ED] (column-major k x q):
// Cp[(g * nt + t) * 64 + lane] = A-operand value of lane (kk = lane>>4, n = lane&15) of group g for k-step t.
__global__ __launch_bounds__(256) void pack_coef_mfma(const double *__restrict__ C, int64_t ldc, int k, int q, int cplx,
                                                      double sign, double *__restrict__ Cp) {
    const int QB = cplx ? 8 : 16, ED = cplx ? 2 : 1;
    const int ngroups = (q + QB - 1) / QB, nt = (k + 3) >> 2;
    const int total = ngroups * nt * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, t = (idx >> 6) % nt, g = idx / (64 * nt);
        const int kk = lane >> 4, nn = lane & 15;
        const int col = 4 * t + kk;
        double v = 0.0;
        if (col < k) {
            if (!cplx) {
                const int qq = g * QB + nn;
                if (qq < q) v = sign * C[(int64_t)qq * ldc + col];
            } else {
                const int qq = g * QB + (nn & 7);                  // [Cr | Ci]
                if (qq < q) v = sign * C[((int64_t)qq * ldc + col) * ED + (nn < 8 ? 0 : 1)];
            }
        }
        Cp[idx] = v;
    }
}

// repack device coefficients laid out [q][ldc][ED] (column-major k x q block, e.g. the sections finish_partials
// leaves in the reduction buffer) into panel_gemm's [group][j][qq][ED] layout with a sign and zero padding.
__global__ __launch_bounds__(256) void pack_coef(const double *__restrict__ C, int64_t ldc, int k, int q, int QB, int ED,
                                                 double sign, double *__restrict__ Cp) {
    const int ngroups = (q + QB - 1) / QB;
    const int total = ngroups * k * QB * ED;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx % ED, qq = (idx / ED) % QB, j = (idx / (ED * QB)) % k, g = idx / (ED * QB * k);
        const int qcol = g * QB + qq;
        Cp[idx] = (qcol < q) ? sign * C[((int64_t)qcol * ldc + j) * ED + e] : 0.0;
    }
}

// out[s] = sum_b partial[s*pstride + b], b < nblocks, fixed order: one wave per slot.
__global__ __launch_bounds__(256) void finish_partials(const double *__restrict__ partial, int64_t pstride,
                                                       int nblocks, int nslots, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= nslots) return;
    const double *p = partial + (int64_t)slot * pstride;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += p[b];
    s = wave_sum(s);
    if (lane == 0) out[slot] = s;
}

// =====================================================================================
// BLAS-1 (abstract_vector TBPs).  Grid-stride, 16 B per lane.
// Shape from tools/blas1_probe.hip (n = 1e8, profiles/r02_blas1_probe.txt): these two- and three-stream kernels run
// fastest with FEW loads in flight -- 2 blocks of 256 threads per CU, two independent 16-byte accesses per lane for the
// 1- and 2-stream reads (scal, dot), one per stream for axpby -- and with non-temporal accesses: 6.2 / 6.5 / 7.2 TB/s
// for scal / axpby / dot against 4.7 / 4.9 / 5.6 at 8 blocks per CU with plain accesses; more blocks or deeper
// unrolling LOSE bandwidth.  `nt` (block-uniform) is set by the launcher for vectors that do not fit the caches anyway.
// =====================================================================================
template <bool CPLX>
__global__ __launch_bounds__(256) void k_scal(double *__restrict__ x, int64_t n, double ar, double ai,
                                              const double *__restrict__ inv_sqrt_of, double tol, Guard guard,
                                              int *__restrict__ stop_out, double tol_break, int nt) {
    // inv_sqrt_of != NULL: alpha = 1/sqrt(|*inv_sqrt_of|) read on the device (fused normalise);
    // skipped (alpha = 1) when the norm is below tol so the host can take the breakdown path.
    // stop_out != NULL (asynchronous Arnoldi): a norm below tol_break, or a NaN, stops every LATER step.
    if (stopped(guard)) return;
    if (inv_sqrt_of) {
        const double nr = sqrt(fabs(*inv_sqrt_of));
        if (stop_out && blockIdx.x == 0 && threadIdx.x == 0 && !(nr >= tol_break)) *stop_out = guard.step;
        if (!(nr >= tol)) return;
        ar = 1.0 / nr;
        ai = 0.0;
    }
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    const int64_t nd = n * ED, nv = nd / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    v2d *xv = reinterpret_cast<v2d *>(x);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto f = [&](v2d v) { if constexpr (CPLX) return cmul(v2d{ar, ai}, v); else return v * ar; };
    if (nt) {
        for (; i + stride < nv; i += 2 * stride) {
            const v2d v0 = __builtin_nontemporal_load(xv + i), v1 = __builtin_nontemporal_load(xv + i + stride);
            __builtin_nontemporal_store(f(v0), xv + i);
            __builtin_nontemporal_store(f(v1), xv + i + stride);
        }
    } else {
        for (; i + stride < nv; i += 2 * stride) {
            const v2d v0 = xv[i], v1 = xv[i + stride];
            xv[i] = f(v0);
            xv[i + stride] = f(v1);
        }
    }
    for (; i < nv; i += stride) xv[i] = f(xv[i]);
    if (!CPLX && (nd & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[nd - 1] *= ar;
}

template <bool CPLX>
__global__ __launch_bounds__(256) void k_axpby(double ar, double ai, const double *__restrict__ x, double br,
                                               double bi, double *__restrict__ y, int64_t n, int nt) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    const int64_t nd = n * ED, nv = nd / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const v2d *xv = reinterpret_cast<const v2d *>(x);
    v2d *yv = reinterpret_cast<v2d *>(y);
    const bool bzero = (br == 0.0 && bi == 0.0);
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto ax = [&](v2d a) { if constexpr (CPLX) return cmul(v2d{ar, ai}, a); else return a * ar; };
    auto by = [&](v2d b) { if constexpr (CPLX) return cmul(v2d{br, bi}, b); else return b * br; };
    // four loops so that both loads of an element are issued back to back, with no data-dependent branch between them
    if (bzero) {                  // one read stream + one write stream: two elements per lane in flight, like k_scal
        int64_t i = i0;
        if (nt) {
            for (; i + stride < nv; i += 2 * stride) {
                const v2d a0 = __builtin_nontemporal_load(xv + i), a1 = __builtin_nontemporal_load(xv + i + stride);
                __builtin_nontemporal_store(ax(a0), yv + i);
                __builtin_nontemporal_store(ax(a1), yv + i + stride);
            }
        } else {
            for (; i + stride < nv; i += 2 * stride) {
                const v2d a0 = xv[i], a1 = xv[i + stride];
                yv[i] = ax(a0);
                yv[i + stride] = ax(a1);
            }
        }
        for (; i < nv; i += stride) yv[i] = ax(xv[i]);
    } else if (nt) {
        for (int64_t i = i0; i < nv; i += stride) {
            const v2d a = __builtin_nontemporal_load(xv + i), b = __builtin_nontemporal_load(yv + i);
            v2d r = ax(a);
            r += by(b);
            __builtin_nontemporal_store(r, yv + i);
        }
    } else {
        for (int64_t i = i0; i < nv; i += stride) {
            const v2d a = xv[i], b = yv[i];
            v2d r = ax(a);
            r += by(b);
            yv[i] = r;
        }
    }
    if (!CPLX && (nd & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        double r = ar * x[nd - 1];
        if (!bzero) r += br * y[nd - 1];
        y[nd - 1] = r;
    }
}

// copy(out, from): nd doubles, same shape as the beta == 0 branch above without the multiply (bit copy)
__global__ __launch_bounds__(256) void k_copy(const double *__restrict__ x, double *__restrict__ y, int64_t nd, int nt) {
    const int64_t nv = nd / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const v2d *xv = reinterpret_cast<const v2d *>(x);
    v2d *yv = reinterpret_cast<v2d *>(y);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nt) {
        for (; i + stride < nv; i += 2 * stride) {
            const v2d a0 = __builtin_nontemporal_load(xv + i), a1 = __builtin_nontemporal_load(xv + i + stride);
            __builtin_nontemporal_store(a0, yv + i);
            __builtin_nontemporal_store(a1, yv + i + stride);
        }
    }
    for (; i < nv; i += stride) yv[i] = xv[i];
    if ((nd & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[nd - 1] = x[nd - 1];
}

// partial[0*pstride + b] (+ [1*pstride + b] imag) = sum conj(x) y over this block's rows
template <bool CPLX>
__global__ __launch_bounds__(256) void k_dot(const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                                             double *__restrict__ partial, int64_t pstride, int nt) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    const int64_t nd = n * ED, nv = nd / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const v2d *xv = reinterpret_cast<const v2d *>(x);
    const v2d *yv = reinterpret_cast<const v2d *>(y);
    v2d acc = v2d{0.0, 0.0};
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto add = [&](v2d a, v2d b) { if constexpr (CPLX) acc += cmulconj(a, b); else acc += a * b; };
    if (x == y) {                 // norm: ONE stream, so four loads per lane in flight
        for (; i + 3 * stride < nv; i += 4 * stride) {
            v2d a0, a1, a2, a3;
            if (nt) {
                a0 = __builtin_nontemporal_load(xv + i); a1 = __builtin_nontemporal_load(xv + i + stride);
                a2 = __builtin_nontemporal_load(xv + i + 2 * stride); a3 = __builtin_nontemporal_load(xv + i + 3 * stride);
            } else {
                a0 = xv[i]; a1 = xv[i + stride]; a2 = xv[i + 2 * stride]; a3 = xv[i + 3 * stride];
            }
            add(a0, a0); add(a1, a1); add(a2, a2); add(a3, a3);
        }
    } else if (nt) {
        for (; i + stride < nv; i += 2 * stride) {
            const v2d a0 = __builtin_nontemporal_load(xv + i), b0 = __builtin_nontemporal_load(yv + i);
            const v2d a1 = __builtin_nontemporal_load(xv + i + stride), b1 = __builtin_nontemporal_load(yv + i + stride);
            add(a0, b0); add(a1, b1);
        }
    } else {
        for (; i + stride < nv; i += 2 * stride) {
            const v2d a0 = xv[i], b0 = yv[i], a1 = xv[i + stride], b1 = yv[i + stride];
            add(a0, b0); add(a1, b1);
        }
    }
    for (; i < nv; i += stride) add(xv[i], yv[i]);
    if (!CPLX && (nd & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc.x += x[nd - 1] * y[nd - 1];
    __shared__ double red[2 * 4];
    double re, im = 0.0;
    if constexpr (CPLX) { re = wave_sum(acc.x); im = wave_sum(acc.y); }
    else re = wave_sum(acc.x + acc.y);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave] = re; red[4 + wave] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        partial[pstride + blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

template <bool CPLX>
__global__ __launch_bounds__(256) void k_rand(double *__restrict__ x, int64_t n, uint64_t seed, int64_t row0) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t g = (uint64_t)(row0 + i);
        if constexpr (CPLX) {
            x[2 * i] = 2.0 * u01(seed, 2 * g) - 1.0;
            x[2 * i + 1] = 2.0 * u01(seed, 2 * g + 1) - 1.0;
        } else {
            x[i] = 2.0 * u01(seed, g) - 1.0;
        }
    }
}

// =====================================================================================
// Operators
// =====================================================================================
template <bool CPLX>
__global__ __launch_bounds__(256) void k_diag(const double *__restrict__ d, const double *__restrict__ x,
                                              double *__restrict__ y, int64_t n, int conj_d, Guard guard) {
    if (stopped(guard)) return;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    const int64_t nd = n * ED, nv = nd / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const v2d *dv = reinterpret_cast<const v2d *>(d);
    const v2d *xv = reinterpret_cast<const v2d *>(x);
    v2d *yv = reinterpret_cast<v2d *>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        v2d a = dv[i], b = xv[i];
        if constexpr (CPLX) yv[i] = conj_d ? cmulconj(a, b) : cmul(a, b);
        else yv[i] = a * b;
    }
    if (!CPLX && (nd & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[nd - 1] = d[nd - 1] * x[nd - 1];
}

__global__ __launch_bounds__(256) void k_diag_linspace(double d0, double dstep, int64_t row0,
                                                       const double *__restrict__ x, double *__restrict__ y,
                                                       int64_t n, Guard guard) {
    if (stopped(guard)) return;
    const int64_t nv = n / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const v2d *xv = reinterpret_cast<const v2d *>(x);
    v2d *yv = reinterpret_cast<v2d *>(y);
    auto dd = [&](int64_t i) {
        const double g = (double)(row0 + 2 * i);
        return v2d{fma(dstep, g, d0), fma(dstep, g + 1.0, d0)};     // ONE rounding per d_i
    };
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < nv; i += 2 * stride) {                      // two elements per lane in flight (one read + one write stream)
        const v2d a0 = __builtin_nontemporal_load(xv + i), a1 = __builtin_nontemporal_load(xv + i + stride);
        yv[i] = dd(i) * a0;
        yv[i + stride] = dd(i + stride) * a1;
    }
    for (; i < nv; i += stride) yv[i] = dd(i) * xv[i];
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = fma(dstep, (double)(row0 + n - 1), d0) * x[n - 1];
}

// y = A x, A in CSR (0-based, int64 row pointers, int32 column indices).  W lanes per row (W = the power of two next to
// the mean row length, 2..64): lane l of a row's group sums entries l, l + W, ... in index order, the W partial sums meet
// in a fixed xor tree (deterministic).  Consecutive groups read consecutive rows, so short rows still coalesce.
// x in two pieces (row-sharded operator with a COMPRESSED exchange): column indices below `nloc` address this rank's own rows in
// `x`, the others the packed entries received from the other ranks in `xrem`; a whole x is (x, anything, nloc = INT64_MAX).
template <bool CPLX>
__device__ __forceinline__ const double *csr_x(const double *__restrict__ x, const double *__restrict__ xrem, int64_t nloc, int64_t j) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    return j < nloc ? x + j * ED : xrem + (j - nloc) * ED;
}

template <bool CPLX, int W>
__global__ __launch_bounds__(256) void k_csr(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colind,
                                             const double *__restrict__ vals, const double *__restrict__ x,
                                             double *__restrict__ y, int64_t n, Guard guard,
                                             const double *__restrict__ xrem, int64_t nloc) {
    if (stopped(guard)) return;
    const int sub = threadIdx.x % W;
    const int64_t groups = (int64_t)gridDim.x * (256 / W);
    for (int64_t r = (int64_t)blockIdx.x * (256 / W) + threadIdx.x / W; r < n; r += groups) {
        double sr = 0.0, si = 0.0;
        const int64_t p1 = rowptr[r + 1];
        for (int64_t p = rowptr[r] + sub; p < p1; p += W) {
            const int64_t j = colind[p];
            const double *xp = csr_x<CPLX>(x, xrem, nloc, j);
            if constexpr (CPLX) {
                const v2d a = *reinterpret_cast<const v2d *>(vals + 2 * p);
                const v2d b = *reinterpret_cast<const v2d *>(xp);
                sr += a.x * b.x - a.y * b.y;
                si += a.x * b.y + a.y * b.x;
            } else {
                sr += vals[p] * xp[0];
            }
        }
#pragma unroll
        for (int off = W / 2; off > 0; off >>= 1) {              // the W lanes of a row share r: all of them are here
            sr += __shfl_xor(sr, off, 64);
            if constexpr (CPLX) si += __shfl_xor(si, off, 64);
        }
        if (sub == 0) {
            if constexpr (CPLX) *reinterpret_cast<v2d *>(y + 2 * r) = v2d{sr, si};
            else y[r] = sr;
        }
    }
}

// CSR product for SHORT rows ("CSR-stream"): a block owns a run of consecutive rows holding at most CSR_NNZ entries
// (rowblocks[], built on the host); its 256 threads stream those entries -- values and column indices fully coalesced,
// x gathered -- into LDS as products, then one thread per row adds its segment in index order (the order a sequential
// host loop uses).  A row longer than CSR_NNZ is a block of its own: strided partial sums, then a fixed tree.
constexpr int CSR_NNZ = 2048;
template <bool CPLX>
__global__ __launch_bounds__(256) void k_csr_stream(const int64_t *__restrict__ rowblocks, const int64_t *__restrict__ rowptr,
                                                    const int32_t *__restrict__ colind, const double *__restrict__ vals,
                                                    const double *__restrict__ x, double *__restrict__ y, int64_t nblocks, Guard guard,
                                                    const double *__restrict__ xrem, int64_t nloc) {
    if (stopped(guard)) return;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    __shared__ double prod[CSR_NNZ * ED];
    __shared__ double red[2 * 4];
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int64_t r0 = rowblocks[b], r1 = rowblocks[b + 1];
        const int64_t p0 = rowptr[r0], p1 = rowptr[r1];
        const int64_t nnzb = p1 - p0;
        if (nnzb > CSR_NNZ) {                                   // one long row
            double sr = 0.0, si = 0.0;
            for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
                const int64_t j = colind[p];
                const double *xp = csr_x<CPLX>(x, xrem, nloc, j);
                if constexpr (CPLX) {
                    const v2d a = *reinterpret_cast<const v2d *>(vals + 2 * p), c = *reinterpret_cast<const v2d *>(xp);
                    sr += a.x * c.x - a.y * c.y;
                    si += a.x * c.y + a.y * c.x;
                } else {
                    sr += vals[p] * xp[0];
                }
            }
            sr = wave_sum(sr);
            if constexpr (CPLX) si = wave_sum(si);
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            __syncthreads();
            if (lane == 0) { red[wave] = sr; red[4 + wave] = si; }
            __syncthreads();
            if (threadIdx.x == 0) {
                y[r0 * ED] = (red[0] + red[1]) + (red[2] + red[3]);
                if constexpr (CPLX) y[r0 * ED + 1] = (red[4] + red[5]) + (red[6] + red[7]);
            }
            continue;
        }
        __syncthreads();                                        // the previous block's segments have been read
        for (int64_t i = threadIdx.x; i < nnzb; i += 256) {
            const int64_t p = p0 + i, j = colind[p];
            const double *xp = csr_x<CPLX>(x, xrem, nloc, j);
            if constexpr (CPLX) {
                const v2d a = *reinterpret_cast<const v2d *>(vals + 2 * p), c = *reinterpret_cast<const v2d *>(xp);
                prod[2 * i] = a.x * c.x - a.y * c.y;
                prod[2 * i + 1] = a.x * c.y + a.y * c.x;
            } else {
                prod[i] = vals[p] * xp[0];
            }
        }
        __syncthreads();
        for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
            const int64_t q0 = rowptr[r] - p0, q1 = rowptr[r + 1] - p0;
            double sr = 0.0, si = 0.0;
            for (int64_t q = q0; q < q1; ++q) {
                if constexpr (CPLX) { sr += prod[2 * q]; si += prod[2 * q + 1]; }
                else sr += prod[q];
            }
            if constexpr (CPLX) *reinterpret_cast<v2d *>(y + 2 * r) = v2d{sr, si};
            else y[r] = sr;
        }
    }
}

// Dense operator (dense_linop, AbstractLinops.fypp:608-660): A is nrows x ncols column-major on the device, leading dimension
// lda EVEN for the real kind and the base 16-byte aligned (the launcher pads), so that a lane reads 16 bytes of a column.
// Square on one rank; the row-sharded operator holds the row block of a rank (nrows = n_local, ncols = n_global).
//
// y = A x: lanes along rows (16 B per lane: a block of 256 threads owns 512 real / 256 complex rows), the columns of the
// block's chunk (split over gridDim.y so that short matrices still fill the chip) walked 8 at a time -- 8 independent
// non-temporal loads per lane in flight, x(j) a wave-uniform scalar load.  Chunk c writes its partial sums to
// part[c][row]; k_gemv_n_finish adds the chunks in index order (deterministic: no atomics).
template <bool CPLX>
__global__ __launch_bounds__(256) void k_gemv_n(const double *__restrict__ A, int64_t lda, int64_t nrows, int64_t ncols,
                                                const double *__restrict__ x, double *__restrict__ part, int64_t cols_per_chunk,
                                                Guard guard) {
    if (stopped(guard)) return;
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int U = 8;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ROWS;
    const int64_t j0 = (int64_t)blockIdx.y * cols_per_chunk;
    int64_t j1 = j0 + cols_per_chunk;
    j1 = j1 < ncols ? j1 : ncols;
    const bool full = r + ROWS <= nrows;
    const int64_t cs = lda * ED;
    v2d acc = v2d{0.0, 0.0};
    if (r < nrows) {
        const double *__restrict__ Ar = A + r * ED;
        int64_t j = j0;
        if (full) {
            for (; j + U <= j1; j += U) {
                v2d a[U];
#pragma unroll
                for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(Ar + (j + u) * cs));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if constexpr (CPLX) acc += cmul(a[u], v2d{x[2 * (j + u)], x[2 * (j + u) + 1]});
                    else acc += a[u] * x[j + u];
                }
            }
        }
        for (; j < j1; ++j) {
            const v2d a = load_y<CPLX>(A + j * cs, r, nrows, full);
            if constexpr (CPLX) acc += cmul(a, v2d{x[2 * j], x[2 * j + 1]});
            else acc += a * x[j];
        }
        store_rows<CPLX>(part + (int64_t)blockIdx.y * nrows * ED, r, nrows, full, acc);
    }
}

// y(i) = sum over chunks of part[c][i] in index order
__global__ __launch_bounds__(256) void k_gemv_n_finish(const double *__restrict__ part, int64_t nd, int nchunks, double *__restrict__ y,
                                                       Guard guard) {
    if (stopped(guard)) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * nd + i];
        y[i] = s;
    }
}

// y = A^H x (y: ncols entries, x: nrows): one wave per column, lanes along rows with 16-byte loads, four in flight; x comes
// from the caches (every wave re-reads it).  DPP wave sum: fixed order.
template <bool CPLX>
__global__ __launch_bounds__(256) void k_gemv_h(const double *__restrict__ A, int64_t lda, int64_t nrows, int64_t ncols,
                                                const double *__restrict__ x, double *__restrict__ y, Guard guard) {
    if (stopped(guard)) return;
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int U = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * 4 + wave;
    if (j >= ncols) return;                       // whole waves leave: the DPP sum below sees full waves only
    const double *__restrict__ Aj = A + j * lda * ED;
    v2d acc = v2d{0.0, 0.0};
    const int64_t step = 64 * ROWS;
    int64_t r = (int64_t)lane * ROWS;
    const int64_t nfull = nrows - (nrows % ROWS);
    for (; r + (U - 1) * step + ROWS <= nfull; r += U * step) {
        v2d a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(Aj + (r + u * step) * ED));
            b[u] = *reinterpret_cast<const v2d *>(x + (r + u * step) * ED);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (CPLX) acc += cmulconj(a[u], b[u]);
            else acc += a[u] * b[u];
        }
    }
    for (; r < nrows; r += step) {
        const v2d a = load_y<CPLX>(Aj, r, nrows, false), b = load_y<CPLX>(x, r, nrows, false);
        if constexpr (CPLX) acc += cmulconj(a, b);
        else acc += a * b;
    }
    if constexpr (CPLX) {
        const double re = wave_sum(acc.x), im = wave_sum(acc.y);
        if (lane == 0) { y[2 * j] = re; y[2 * j + 1] = im; }
    } else {
        const double re = wave_sum(acc.x + acc.y);
        if (lane == 0) y[j] = re;
    }
}

// y(0:nd) = z(0:nd) (doubles) under the asynchronous pipeline's guard: the slice of an all-reduced full-length vector that
// this rank owns (row-sharded rmatvec)
__global__ __launch_bounds__(256) void k_copy_guarded(const double *__restrict__ z, double *__restrict__ y, int64_t nd, Guard guard) {
    if (stopped(guard)) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += (int64_t)gridDim.x * blockDim.x) y[i] = z[i];
}

// out[i] = x[idx[i]] (ED doubles each): the entries of this rank's block that other ranks' rows reference, packed for the
// compressed exchange of the row-sharded CSR operator
template <bool CPLX>
__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ x, const int32_t *__restrict__ idx, int64_t cnt,
                                              double *__restrict__ out, Guard guard) {
    if (stopped(guard)) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
        if constexpr (CPLX) reinterpret_cast<v2d *>(out)[i] = reinterpret_cast<const v2d *>(x)[idx[i]];
        else out[i] = x[idx[i]];
    }
}

// 5-point Laplacian, N x N grid, Dirichlet, scale s = (N+1)^2.  One thread per 2 grid points
// along the fast index; neighbours come from L1/L2 (each row is re-used by 3 stencil rows).
// Row-sharded: this rank holds NJ consecutive grid lines; `lo` / `hi` are the neighbouring ranks' boundary lines
// (N doubles each, delivered by the halo exchange) or NULL where the global Dirichlet boundary is.
__global__ __launch_bounds__(256) void k_lap5(const double *__restrict__ u, double *__restrict__ v, int64_t N, int64_t NJ,
                                              const double *__restrict__ lo, const double *__restrict__ hi, double s,
                                              Guard guard) {
    if (stopped(guard)) return;
    if ((N & 1) == 0) {
        // even N (every lane owns a 16-byte aligned pair): centre / upper / lower pairs are ONE 16-byte load each, the left
        // and right neighbours come from the adjacent lanes' centre pairs (wave shuffle; only the wave's edge lanes load
        // them) -- 3 loads per 2 points instead of 8.  Same operations in the same order as the generic path below.
        // Persistent blocks walk (grid line, 512-point segment) tiles in storage order: gridDim.y == 1 on this path.
        const int64_t nseg = (N / 2 + 255) / 256;
        for (int64_t t = blockIdx.x; t < NJ * nseg; t += gridDim.x) {
        const int64_t j = t / nseg;
        const int64_t i = ((t - j * nseg) * 256 + threadIdx.x) * 2;
        const bool act = i < N;
        const int64_t c = i + j * N;
        v2d cc = v2d{0.0, 0.0}, up = cc, dn = cc;
        if (act) {
            cc = *reinterpret_cast<const v2d *>(u + c);
            if (j > 0) up = *reinterpret_cast<const v2d *>(u + c - N);
            else if (lo) up = *reinterpret_cast<const v2d *>(lo + i);
            if (j < NJ - 1) dn = *reinterpret_cast<const v2d *>(u + c + N);
            else if (hi) dn = *reinterpret_cast<const v2d *>(hi + i);
        }
        const int lane = threadIdx.x & 63;
        double left = __shfl_up(cc.y, 1, 64), right = __shfl_down(cc.x, 1, 64);
        if (act && lane == 0 && i > 0) left = u[c - 1];
        if (act && lane == 63 && i + 2 < N) right = u[c + 2];
        if (!act) continue;
        double a0 = 4.0 * cc.x, a1 = 4.0 * cc.y;
        if (i > 0) a0 -= left;
        a0 -= cc.y;
        a1 -= cc.x;
        if (i + 2 < N) a1 -= right;
        if (j > 0 || lo) { a0 -= up.x; a1 -= up.y; }
        if (j < NJ - 1 || hi) { a0 -= dn.x; a1 -= dn.y; }
        *reinterpret_cast<v2d *>(v + c) = v2d{s * a0, s * a1};
        }
        return;
    }
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int64_t j = blockIdx.y;
    if (i >= N) return;
    const int64_t c = i + j * N;
    const bool two = (i + 1 < N);
    double c0 = u[c], c1 = two ? u[c + 1] : 0.0;
    double a0 = 4.0 * c0, a1 = 4.0 * c1;
    if (i > 0) a0 -= u[c - 1];
    a0 -= two ? c1 : 0.0;
    a1 -= c0;
    if (i + 2 < N) a1 -= u[c + 2];
    if (j > 0) { a0 -= u[c - N]; if (two) a1 -= u[c + 1 - N]; }
    else if (lo) { a0 -= lo[i]; if (two) a1 -= lo[i + 1]; }
    if (j < NJ - 1) { a0 -= u[c + N]; if (two) a1 -= u[c + 1 + N]; }
    else if (hi) { a0 -= hi[i]; if (two) a1 -= hi[i + 1]; }
    v[c] = s * a0;
    if (two) v[c + 1] = s * a1;
}

// Linearised complex Ginzburg-Landau right-hand side, the reference example's stencil INCLUDING its
// boundary rows (example/ginzburg_landau/Ginzburg_Landau.f90:126-136; adjoint :170-179):
//   f(v)_i = -nu*cu + gamma*d2u + mu_i v_i,   mu_i = mu_c + (mu2/2) x_i^2,  x_i = -L/2 + i*dx (i = 1..n).
// One classical RK4 stage per launch: v = u + a*kprev (formed on the fly at the 3 stencil points),
// kout = f(v), acc (+)= b*kout (first stage: acc = u + b*kout).
__global__ __launch_bounds__(256) void k_gl_stage(const double *__restrict__ u, const double *__restrict__ kprev,
                                                  double a, double *__restrict__ kout, double *__restrict__ acc,
                                                  double b, int first, int64_t n, double dx, double halfL,
                                                  double nu_re, double nu_im, double ga_re, double ga_im,
                                                  double mu_c, double mu2, int adjoint, int64_t row0, int64_t n_global,
                                                  const double *__restrict__ halo, Guard guard) {
    // n = LOCAL rows [row0, row0 + n) of n_global; halo = {v(row0 - 1), v(row0 + n)} (2 complex) from the neighbouring
    // ranks when the block does not touch the global boundary (NULL for a single rank).
    if (stopped(guard)) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // 0-based local; reference index row0 + i + 1
    if (i >= n) return;
    const int64_t gi = row0 + i;
    const v2d *uv = reinterpret_cast<const v2d *>(u);
    const v2d *kv = reinterpret_cast<const v2d *>(kprev);
    const v2d *hv = reinterpret_cast<const v2d *>(halo);
    auto val = [&](int64_t j) -> v2d {
        if (j < 0) return hv[0];
        if (j >= n) return hv[1];
        v2d r = uv[j];
        if (kprev) r += kv[j] * a;
        return r;
    };
    const v2d c = val(i);
    v2d cu, d2u;
    const double inv2dx = 1.0 / (2.0 * dx), invdx2 = 1.0 / (dx * dx);
    if (n_global == 1) {
        cu = v2d{0.0, 0.0};
        d2u = c * (-2.0) * invdx2;
    } else if (gi == 0) {
        const v2d r = val(i + 1);
        cu = r * inv2dx;
        d2u = (r - c * 2.0) * invdx2;
    } else if (gi == n_global - 1) {
        const v2d l = val(i - 1);
        cu = l * (-inv2dx);
        d2u = (c * (-2.0) + l) * inv2dx;          // sic: the reference divides this row by 2*dx
    } else {
        const v2d l = val(i - 1), r = val(i + 1);
        cu = (r - l) * inv2dx;
        d2u = (r - c * 2.0 + l) * invdx2;
    }
    const double x = -halfL + (double)(gi + 1) * dx;
    const double mu = mu_c + 0.5 * mu2 * x * x;
    v2d f;
    if (adjoint) f = cmul(v2d{nu_re, -nu_im}, cu) + cmul(v2d{ga_re, -ga_im}, d2u) + c * mu;
    else f = cmul(v2d{-nu_re, -nu_im}, cu) + cmul(v2d{ga_re, ga_im}, d2u) + c * mu;
    reinterpret_cast<v2d *>(kout)[i] = f;
    v2d *av = reinterpret_cast<v2d *>(acc);
    av[i] = (first ? uv[i] : av[i]) + f * b;
}

// the two edge values v = u + a*kprev of this rank's block, for the halo exchange of the next stage
__global__ void k_gl_edges(const double *__restrict__ u, const double *__restrict__ kprev, double a, int64_t n,
                           double *__restrict__ sendbuf, Guard guard) {
    if (stopped(guard)) return;
    if (threadIdx.x >= 2) return;
    const int64_t j = threadIdx.x == 0 ? 0 : n - 1;
    v2d r = reinterpret_cast<const v2d *>(u)[j];
    if (kprev) r += reinterpret_cast<const v2d *>(kprev)[j] * a;
    reinterpret_cast<v2d *>(sendbuf)[threadIdx.x] = r;
}

}  // namespace lk
