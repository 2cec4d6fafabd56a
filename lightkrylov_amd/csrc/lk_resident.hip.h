// lk_resident.hip.h -- the WHOLE double Gram-Schmidt step of one vector as ONE persistent launch (round 6).
//
// For panels that fit the 256 MB memory-side cache the three-sweep schedule of lk_kernels.hip.h is bound by its launches, not by
// bytes: three sweeps + three finish kernels + the normalise = seven kernel boundaries around 10-20 us of streaming
// (profiles/r05_dgs_size_scan.jsonl: n = 3 10^5, k = 32 -> 96 us per step at 2.5 TB/s).  Here every block OWNS a contiguous run of
// row tiles for the whole step and walks it once per phase
//   phase 1   h1 = X^H y , ||y||^2                              (gram_schmidt.fypp:126, 141)
//   phase 2   y' = y - X h1 in registers ; h2 = X^H y' , ||y'||^2      (:144-145, second pass :126, 141)
//   phase 3   y'' = (y - X h1) - X h2 stored ; ||y''||^2       (:144-145; y' re-formed exactly as phase 2 formed it)
//   phase 4   y'' <- y'' / ||y''||  (qr_no_pivoting's scale, qr.fypp:165) with the breakdown test of the asynchronous batch
// with a grid-wide SUM between the phases: block partials -> group sums (the blocks b, b + 8, ... that share an XCD under
// round-robin placement: their hand-off stays in one L2; placement is speed only, every access below is agent scope) -> totals,
// every level added in a FIXED order (deterministic, no floating-point atomics).  The second and third walk of a block's rows are
// served from the XCD's L2 / the memory-side cache as far as the panel fits them; phase 2 walks the tiles backwards so that it
// starts on the rows phase 1 touched last.
//
// Hand-off protocol (MI355X_MICROARCH.md, "inter-workgroup visibility", table row 1): payload written with agent-scope (sc1,
// write-through) 8-byte stores, every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds to an agent-scope
// counter; the block whose add came last reduces its group and adds to the top counter; ONE lane per block polls the top counter
// with agent-scope loads (+ s_sleep), joins a workgroup barrier, and every payload load is an agent-scope (sc1) load.
// All blocks must be co-resident: the grid is at most one block per CU.  A spin that outlasts its deadline (another persistent
// kernel holding the CUs) raises the abort word: nothing has been written to y at that point (the first wait comes before any
// store), the launcher falls back to the three-sweep schedule and resets the counters.
#pragma once
#include "lk_kernels.hip.h"

namespace lk {

constexpr int RES_GROUPS = 8;        // block groups (b % 8)
constexpr int RES_EPISODES = 3;      // grid-wide sums per launch
constexpr int RES_CNT_STRIDE = 32;   // unsigneds between two counters: a 128-byte line each
constexpr int RES_NCNT = RES_EPISODES * (RES_GROUPS + 1) + 2;   // per episode: 8 group counters + top ; exit counter ; abort word
constexpr int RES_EXIT = RES_EPISODES * (RES_GROUPS + 1);
constexpr int RES_ABORT = RES_EXIT + 1;

struct ResidentWs {
    double *part;     // [RES_EPISODES][grid][S]        block partials
    double *xsum;     // [RES_EPISODES][RES_GROUPS][S]  group sums
    unsigned *cnt;    // [RES_NCNT][RES_CNT_STRIDE]     all zero between launches (the last block to leave clears them)
    int S;            // slot stride (>= (k + 1) * ED)
};

__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned *res_ctr(const ResidentWs &ws, int i) { return ws.cnt + (size_t)i * RES_CNT_STRIDE; }

// Sum over the grid of this block's partials mine[first, nslots) (LDS) -> tot[first, nslots) (LDS), identical bits in every block.
// Returns false when the launch was aborted (block-uniform).  `ctl` = 2 ints of LDS.
template <int NT>
__device__ __forceinline__ bool grid_sum(const ResidentWs &ws, int ep, int first, int nslots, const double *mine, double *tot, int *ctl,
                                         long long deadline) {
    const int b = blockIdx.x, G = gridDim.x;
    const int g = b % RES_GROUPS;
    const int ngroups = G < RES_GROUPS ? G : RES_GROUPS;
    const int members = (G - g + RES_GROUPS - 1) / RES_GROUPS;       // blocks g, g + 8, ... < G
    double *part = ws.part + ((size_t)ep * G) * ws.S;
    double *xs = ws.xsum + ((size_t)ep * RES_GROUPS) * ws.S;
    for (int o = first + threadIdx.x; o < nslots; o += NT) st_agent(part + (size_t)b * ws.S + o, mine[o]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(res_ctr(ws, ep * (RES_GROUPS + 1) + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[0] = ticket == (unsigned)(members - 1);
    }
    __syncthreads();
    if (ctl[0]) {                                                    // the group's last arrival adds its members in member order
        for (int o = first + threadIdx.x; o < nslots; o += NT) {
            double s = 0.0;
            int m = 0;
            for (; m + 8 <= members; m += 8) {
                double v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = ld_agent(part + (size_t)(g + (m + i) * RES_GROUPS) * ws.S + o);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
            }
            for (; m < members; ++m) s += ld_agent(part + (size_t)(g + m * RES_GROUPS) * ws.S + o);
            st_agent(xs + (size_t)g * ws.S + o, s);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            (void)__hip_atomic_fetch_add(res_ctr(ws, ep * (RES_GROUPS + 1) + RES_GROUPS), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        const unsigned *top = res_ctr(ws, ep * (RES_GROUPS + 1) + RES_GROUPS);
        unsigned *abortw = res_ctr(ws, RES_ABORT);
        int ok = 1;
        for (unsigned it = 0;; ++it) {
            if (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)ngroups) break;
            if ((it & 15) == 15) {
                if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                if (wall_clock64() > deadline) {
                    __hip_atomic_store(abortw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        ctl[1] = ok;
    }
    __syncthreads();
    if (!ctl[1]) return false;
    for (int o = first + threadIdx.x; o < nslots; o += NT) {
        double v[RES_GROUPS];
#pragma unroll
        for (int i = 0; i < RES_GROUPS; ++i) v[i] = i < ngroups ? ld_agent(xs + (size_t)i * ws.S + o) : 0.0;
        double s = v[0];
#pragma unroll
        for (int i = 1; i < RES_GROUPS; ++i) s += v[i];
        tot[o] = s;
    }
    __syncthreads();
    return true;
}

// y as this lane stored it a moment ago: served from L2 (the line the CU's L1 holds predates the store)
template <bool CPLX>
__device__ __forceinline__ v2d load_y_l2(const double *__restrict__ y, int64_t r, int64_t n, bool full) {
    if (full) return __builtin_nontemporal_load(reinterpret_cast<const v2d *>(y + r * K<CPLX>::ELEM_DOUBLES));
    if constexpr (CPLX) {
        return (r < n) ? v2d{ld_agent(y + 2 * r), ld_agent(y + 2 * r + 1)} : v2d{0.0, 0.0};
    } else {
        v2d yv;
        yv.x = (r < n) ? ld_agent(y + r) : 0.0;
        yv.y = (r + 1 < n) ? ld_agent(y + r + 1) : 0.0;
        return yv;
    }
}

template <bool CPLX, int KC, bool NT>
__device__ __forceinline__ void res_load_cols(const double *__restrict__ Xw, int64_t colstride, int64_t r, int64_t n, bool full, int nc,
                                              v2d (&xv)[KC]) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    if (full) {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            if (jj < nc) {
                const v2d *p = reinterpret_cast<const v2d *>(Xw + jj * colstride + r * ED);
                if constexpr (NT) xv[jj] = __builtin_nontemporal_load(p);
                else xv[jj] = *p;
            } else xv[jj] = v2d{0.0, 0.0};
        }
    } else {
        load_cols<CPLX, KC, false>(Xw, colstride, r, n, false, nc, xv);
    }
}

// One phase over the block's tiles [t0, t1): panel_sweep's tile body (SC = 1, G = 1) with the coefficients taken from LDS.
//   MODE 1: dot   MODE 2: update (kept in registers) + dot   MODE 4: two-coefficient update, stored
// Leaves the block's partial sums in mine[0, (k + 1) * ED): slots 0..k-1 = h (MODE 1, 2), slot k = the norm of what y became.
template <bool CPLX, int KC, int NW, bool NT, int MODE>
__device__ __forceinline__ void res_phase(const double *__restrict__ X, int64_t ldx, int k, double *__restrict__ y, int64_t n,
                                          const double *h1, const double *h2, int WC, int kcw, int64_t t0, int64_t t1, bool reverse,
                                          v2d *u_lds, double *red_lds, double *mine) {
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int WROWS = 64 * ROWS;
    constexpr bool UPDATE = MODE != 1, DOT = MODE <= 2, TWO = MODE == 4;
    constexpr int NU = TWO ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave % WC, wr = wave / WC, WR = NW / WC;
    const int c0 = wc * kcw;
    int nc = k - c0;
    nc = nc > kcw ? kcw : nc;
    nc = nc < 0 ? 0 : nc;

    // projection coefficients: read from LDS at every use (all lanes the same address: a broadcast).  Held in registers across the
    // tile loop they would cost 2-4 VGPRs per column and set (they arrive from LDS, not through scalar loads); the index is laundered
    // through an empty asm so that the compiler cannot hoist the reads.  Columns beyond nc read column c0 (finite) against zeros of X.
    auto hcoef = [&](const double *h, int jj) -> v2d {
        int cj = jj < nc ? c0 + jj : 0;
        asm volatile("" : "+v"(cj));
        if constexpr (CPLX) return *reinterpret_cast<const v2d *>(h + 2 * cj);
        else return v2d{h[cj], 0.0};
    };
    v2d acc[DOT ? KC : 1];
#pragma unroll
    for (int jj = 0; jj < (DOT ? KC : 1); ++jj) acc[jj] = v2d{0.0, 0.0};
    double nrm = 0.0;

    const int64_t tile_rows = (int64_t)WR * WROWS;
    const double *Xw = X + (int64_t)c0 * ldx * ED;
    const int64_t colstride = ldx * ED;
    const int64_t roff = (int64_t)wr * WROWS + (int64_t)lane * ROWS;
    int buf = 0;
    for (int64_t i = t0; i < t1; ++i) {
        const int64_t t = reverse ? t1 - 1 - (i - t0) : i;
        const int64_t r = t * tile_rows + roff;
        const bool full = (t + 1) * tile_rows <= n;
        v2d xv[KC];
        v2d yv = load_y<CPLX>(y, r, n, full);
        res_load_cols<CPLX, KC, NT>(Xw, colstride, r, n, full, nc, xv);
        if constexpr (UPDATE) {
            v2d u = v2d{0.0, 0.0}, u2 = v2d{0.0, 0.0};
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if constexpr (CPLX) u += cmul(xv[jj], hcoef(h1, jj));
                else u += xv[jj] * hcoef(h1, jj).x;
            }
            if constexpr (TWO) {
#pragma unroll
                for (int jj = 0; jj < KC; ++jj) {
                    if constexpr (CPLX) u2 += cmul(xv[jj], hcoef(h2, jj));
                    else u2 += xv[jj] * hcoef(h2, jj).x;
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
            if constexpr (TWO) {
                yv -= u2;
                if (wc == 0) store_rows<CPLX>(y, r, n, full, yv, 0);
            }
        }
        if constexpr (DOT) {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if constexpr (CPLX) acc[jj] += cmulconj(xv[jj], yv);
                else acc[jj] += xv[jj] * yv;
            }
        }
        if (wc == 0) nrm += yv.x * yv.x + yv.y * yv.y;
    }

    constexpr int SLOTS = KC * ED + 1;
    if constexpr (DOT) {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
            if constexpr (CPLX) {
                const double re = wave_sum(acc[jj].x), im = wave_sum(acc[jj].y);
                if (lane == 0) { red_lds[wave * SLOTS + 2 * jj] = re; red_lds[wave * SLOTS + 2 * jj + 1] = im; }
            } else {
                const double s = wave_sum(acc[jj].x + acc[jj].y);
                if (lane == 0) red_lds[wave * SLOTS + jj] = s;
            }
        }
    }
    {
        const double s = wave_sum(nrm);
        if (lane == 0) red_lds[wave * SLOTS + KC * ED] = s;
    }
    __syncthreads();
    if constexpr (DOT) {
        for (int o = threadIdx.x; o < k * ED; o += NW * 64) {
            const int j = o / ED, part = o % ED;
            const int cgj = j / kcw, jj = j - cgj * kcw;
            double s = 0.0;
            for (int w = 0; w < WR; ++w) s += red_lds[(w * WC + cgj) * SLOTS + jj * ED + part];
            mine[o] = s;
        }
    }
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < WR; ++w) s += red_lds[(w * WC) * SLOTS + KC * ED];
        mine[k * ED] = s;
        if constexpr (CPLX) mine[k * ED + 1] = 0.0;
    }
    __syncthreads();
}

// flags: bit 0 = normalise y'' (skipped below tol_scale) ; bit 1 = phase 2 walks the tiles backwards
// out: three result sections of `rs` doubles (h1 | ||y||^2 ; h2 | ||y'||^2 ; slot k = ||y''||^2, slot k*ED+1 of the THIRD section =
// launch status: 0 done, 1 given up before anything was written to y, 2 failed after y'' was stored).
template <bool CPLX, int KC, int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void dgs_resident(const double *__restrict__ X, int64_t ldx, int k, double *__restrict__ y, int64_t n,
                                                         ResidentWs ws, double *__restrict__ out, int rs, int WC, int kcw, int flags,
                                                         double tol_scale, double tol_break, int *__restrict__ stop_out,
                                                         long long spin_ticks, Guard guard) {
    if (stopped(guard)) return;
    constexpr int ROWS = K<CPLX>::ROWS;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int NTHR = NW * 64;
    constexpr int CAPS = (KC * NW + 1) * ED;
    __shared__ v2d u_lds[2 * 2 * NW * 64];
    __shared__ double red_lds[NW * (KC * ED + 1)];
    __shared__ __attribute__((aligned(16))) double mine[CAPS + 1], h1[CAPS + 1], h2[CAPS + 1], h3[CAPS + 1];
    __shared__ int ctl[2];
    unsigned *abortw = res_ctr(ws, RES_ABORT);
    if (threadIdx.x == 0) ctl[0] = __hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    __syncthreads();
    if (ctl[0]) return;                                   // a block that became resident after the launch was given up
    __syncthreads();
    // the FIRST wait is the one that can starve (a block not yet resident); past it every block is on the chip.  The later waits
    // still carry a (generous) bound so that a defect can never hang the device; one firing there is reported as a failure.
    const long long start = wall_clock64();
    const long long deadline = start + spin_ticks, deadline_late = deadline + 1000000000ll;
    const int WR = NW / WC;
    const int64_t tile_rows = (int64_t)WR * 64 * ROWS;
    const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    const int64_t t0 = ntiles * blockIdx.x / gridDim.x, t1 = ntiles * (blockIdx.x + 1) / gridDim.x;
    const int nslots = (k + 1) * ED;
    double *r0 = out, *r1 = out + rs, *r2 = out + 2 * (int64_t)rs;
    auto give_up = [&](double status) {
        if (threadIdx.x == 0) {
            r2[k * ED + 1] = status;
            if (stop_out) *stop_out = guard.step;
        }
    };

    res_phase<CPLX, KC, NW, NT, 1>(X, ldx, k, y, n, nullptr, nullptr, WC, kcw, t0, t1, false, u_lds, red_lds, mine);
    if (!grid_sum<NTHR>(ws, 0, 0, nslots, mine, h1, ctl, deadline)) { give_up(1.0); return; }
    res_phase<CPLX, KC, NW, NT, 2>(X, ldx, k, y, n, h1, nullptr, WC, kcw, t0, t1, (flags & 2) != 0, u_lds, red_lds, mine);
    if (!grid_sum<NTHR>(ws, 1, 0, nslots, mine, h2, ctl, deadline_late)) { give_up(1.0); return; }
    res_phase<CPLX, KC, NW, NT, 4>(X, ldx, k, y, n, h1, h2, WC, kcw, t0, t1, false, u_lds, red_lds, mine);
    if (!grid_sum<NTHR>(ws, 2, k * ED, nslots, mine, h3, ctl, deadline_late)) { give_up(2.0); return; }

    const double nr = sqrt(fabs(h3[k * ED]));
    if (blockIdx.x == 0) {
        for (int o = threadIdx.x; o < nslots; o += NTHR) { r0[o] = h1[o]; r1[o] = h2[o]; }
        if (threadIdx.x == 0) {
            r2[k * ED] = h3[k * ED];
            r2[k * ED + 1] = 0.0;
            if (stop_out && !(nr >= tol_break)) *stop_out = guard.step;
        }
    }
    if ((flags & 1) && nr >= tol_scale) {
        // the lanes that stored y'' scale it: same wave, same lane, same address as the store (ordered by the hardware)
        const double inv = 1.0 / nr;
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wc = wave % WC, wr = wave / WC;
        if (wc == 0) {
            for (int64_t t = t0; t < t1; ++t) {
                const int64_t r = t * tile_rows + (int64_t)wr * 64 * ROWS + (int64_t)lane * ROWS;
                const bool full = (t + 1) * tile_rows <= n;
                v2d yv = load_y_l2<CPLX>(y, r, n, full);
                yv *= inv;
                store_rows<CPLX>(y, r, n, full, yv, 0);
            }
        }
    }
    // the last block to leave clears the counters for the next launch (nobody polls any more: every block is past its last wait)
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(res_ctr(ws, RES_EXIT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {
            for (int i = 0; i <= RES_EXIT; ++i) __hip_atomic_store(res_ctr(ws, i), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace lk
