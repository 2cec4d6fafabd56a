// lk_resident.hip.h -- the WHOLE double Gram-Schmidt step of one vector as ONE persistent launch (round 6).
//
// For panels that fit the 256 MB memory-side cache the three-sweep schedule of lk_kernels.hip.h is bound by its launches, not by
// bytes: three sweeps + three finish kernels + the normalise = seven kernel boundaries around 10-20 us of streaming
// (profiles/r05_dgs_size_scan.jsonl: n = 3 10^5, k = 32 -> 96 us per step at 2.5 TB/s).  Here every block OWNS a contiguous run of
// row tiles for the whole step:
//   phase 1   h1 = X^H y , ||y||^2                              (gram_schmidt.fypp:126, 141)
//   phase 2   y' = y - X h1 in registers ; h2 = X^H y' , ||y'||^2      (:144-145, second pass :126, 141)
//   phase 3   y'' = y' - X h2 ; ||y''||^2                       (:144-145)
//   phase 4   y'' <- y'' / ||y''||  (qr_no_pivoting's scale, qr.fypp:165) with the breakdown test of the asynchronous batch
// with a grid-wide SUM between the phases (fixed order at every level: deterministic, no floating-point atomics).
//
// Two kernels:
//   dgs_onchip    the block's tiles of X and y stay IN REGISTERS from phase 1 to the end: X is read from memory ONCE (k + 1 columns
//                 in, one out, against 3k + 5 for the three sweeps), phases 2-4 touch no memory but the sums.  Takes panels up to the
//                 register files' capacity: 32 16-byte values per lane, 512 lanes per CU = 256 KB per CU, 64 MB on the chip.
//   dgs_resident  larger panels that still fit the memory-side cache: the block walks its tiles once per phase (the second and third
//                 walk are served from L2 / the memory-side cache; measured 8-11 TB/s, profiles/r06_resident_phases.jsonl), phase 3
//                 re-forms y' from y as sweep 3 of the three-sweep schedule does.
// In-kernel timeline that shaped them (block 0's clock, first version of dgs_resident): an EMPTY phase cost 4.6 us -- 2 us of it
// the 16 dependent wave_sum chains of the dot epilogue, replaced here by a transpose through LDS --, a grid-wide sum 6 us: seven
// dependent memory round trips (payload drain, ticket, group reduce, drain, top counter, poll, read) -- now two publish-to-seen hops
// of tagged granules (grid_sum below).
//
// Hand-off protocol (MI355X_MICROARCH.md, "inter-workgroup visibility"): 16-byte {value, tag} granules written by ONE agent-scope
// (sc1, write-through) store instruction and read by agent-scope 16-byte loads -- a reader that sees this launch's tag sees the value,
// so nothing has to be drained or counted and nothing is left to clear (the guide's data-tagged granule, R2, with a 64-bit payload).
// All blocks must be co-resident: the grid is at most one block per CU.  A wait that outlasts its deadline (another persistent
// kernel holding part of the chip) raises the abort word: nothing has been written to y at that point (the first wait comes before any
// store), the host runs that step on the three-sweep schedule, clears the word and pauses the single launch for a while (lk_engine.hip).
#pragma once
#include "lk_kernels.hip.h"

namespace lk {

constexpr int RES_GROUPS = 8;        // block groups (b % 8: the blocks that share an XCD under round-robin placement; speed only)
constexpr int RES_EPISODES = 3;      // grid-wide sums per launch
constexpr int RES_GRID_CAP = 512;    // blocks the hand-off buffers hold
constexpr int RES_ABORT = 0;         // word 0 of `cnt`: the launch was given up
constexpr int RES_CNT_STRIDE = 32;
constexpr int RES_ROW = 68;          // doubles per row of the transpose buffer: 64 lanes + 4 (the quarter sums then read conflict-free)

struct ResidentWs {
    unsigned *cnt;    // the abort word (zero between launches unless one gave up: the launcher clears it)
    int S;            // slot stride (>= (k + 1) * ED)
    v2d *gran;        // [RES_EPISODES][RES_GRID_CAP + RES_GROUPS][S]  {value, tag} granules (never cleared: tags are unique per launch)
    unsigned long long epoch;   // this launch's number (> 0)
    long long *tim;   // [8] wall_clock64 of block 0 at the phase boundaries of the last launch (lk_resident_phase_ticks)
};

__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned *res_ctr(const ResidentWs &ws, int i) { return ws.cnt + (size_t)i * RES_CNT_STRIDE; }

// 16-byte agent-scope (sc1) accesses for the {value, tag} granules: one store instruction publishes value and tag together, one load
// reads them together (a naturally aligned 16-byte access is a single request to one cache line).
__device__ __forceinline__ void st16_agent(v2d *p, v2d v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ld16x8_agent(const v2d *const (&p)[8], v2d (&g)[8]) {
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                 : "memory");
}

// Sum over `nrows` rows (row i at src + (row0 + i * rstep) * stride) of the tagged granules of slots [first, first + cnt) -> dst[0, cnt)
// (LDS), in row order.  Thread (run r, slot o) reads rows r, r + R, ... eight per batch, re-reads a batch until every tag is this
// episode's (a reader that sees the tag sees the value: one 16-byte store published both), and adds them in row order; thread o then
// adds the R runs in order.  Returns false when the launch was given up (block-uniform).
template <int NT>
__device__ __forceinline__ bool res_gather(const ResidentWs &ws, const v2d *src, int row0, int rstep, int nrows, int stride, int first, int cnt,
                                           double tagd, double *dst, double *scr, int *ctl, long long deadline) {
    int R = (nrows + 7) / 8;
    if (R > NT / cnt) R = NT / cnt;
    if (R < 1) R = 1;
    const int run = threadIdx.x / cnt, oo = (int)threadIdx.x % cnt;
    const bool active = run < R && cnt <= NT;
    double s = 0.0;
    if (threadIdx.x == 0) ctl[1] = 1;
    for (int i0 = 0; i0 * R < nrows; i0 += 8) {                       // (block-uniform trip count)
        const v2d *p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = run + (i0 + i) * R;
            p[i] = src + (size_t)(row0 + (active && row < nrows ? row : 0) * rstep) * stride + first + (active ? oo : 0);
        }
        v2d g8[8];
        for (unsigned it = 0;; ++it) {
            int ok = 1;
            if (active) {
                ld16x8_agent(p, g8);
#pragma unroll
                for (int i = 0; i < 8; ++i) ok &= __double_as_longlong(g8[i].y) == __double_as_longlong(tagd);
            }
            if (__syncthreads_and(ok)) break;
            if (threadIdx.x == 0) {
                if ((it & 7) == 7 && ld_agent(res_ctr(ws, RES_ABORT)) != 0u) ctl[1] = 0;
                else if (wall_clock64() > deadline) {
                    __hip_atomic_store(res_ctr(ws, RES_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ctl[1] = 0;
                }
            }
            __syncthreads();
            if (!ctl[1]) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s += (run + (i0 + i) * R < nrows) ? g8[i].x : 0.0;
        }
    }
    if (active) scr[threadIdx.x] = s;
    __syncthreads();
    for (int o = threadIdx.x; o < cnt; o += NT) {
        double t = 0.0;
        for (int r = 0; r < R; ++r) t += scr[r * cnt + o];
        dst[o] = t;
    }
    __syncthreads();
    return true;
}

// Sum over the grid of this block's partials mine[first, nslots) (LDS) -> tot[first, nslots) (LDS), identical bits in every block.
// Returns false when the launch was given up (block-uniform).  `ctl` = 2 ints, `scr` = NT doubles of LDS.
// Two levels of {value, tag} granules, tag = (launch number, episode) -- no drain, no counter, nothing to clear afterwards:
//   every block publishes its partials; the LEADER of each block group (blocks 0..7; group g = the blocks g, g + 8, ...) gathers its
//   members' in member order and publishes the group sum; every block gathers the 8 group sums in group order.
// Two publish-to-seen hops on the critical path, G + 8 * G granule reads per slot in all (every block reading every partial
// itself costs G^2: measured slower from 9 slots on).
template <int NT>
__device__ __forceinline__ bool grid_sum(const ResidentWs &ws, int ep, int first, int nslots, const double *mine, double *tot, int *ctl,
                                         double *scr, long long deadline) {
    const int b = blockIdx.x, G = gridDim.x;
    const int ngroups = G < RES_GROUPS ? G : RES_GROUPS;
    const int cnt = nslots - first;
    const double tagd = __longlong_as_double((long long)(ws.epoch * 4ull + (unsigned)ep + 1ull));
    v2d *g0 = ws.gran + ((size_t)ep * (RES_GRID_CAP + RES_GROUPS)) * ws.S;          // [G][S] block partials | [8][S] group sums
    v2d *g1 = g0 + (size_t)RES_GRID_CAP * ws.S;
    for (int o = first + threadIdx.x; o < nslots; o += NT) st16_agent(g0 + (size_t)b * ws.S + o, v2d{mine[o], tagd});
    // one or two slots (the norm of phase 3): every block adds all G partials itself, in block order -- ONE hop, and G^2 granule reads are
    // still only a megabyte (measured 3.0 us against 4.1-4.4 for the two levels)
    if ((int64_t)cnt * G <= 2 * NT) return res_gather<NT>(ws, g0, 0, 1, G, ws.S, first, cnt, tagd, tot + first, scr, ctl, deadline);
    if (b < ngroups) {
        const int members = (G - b + RES_GROUPS - 1) / RES_GROUPS;
        if (!res_gather<NT>(ws, g0, b, RES_GROUPS, members, ws.S, first, cnt, tagd, tot + first, scr, ctl, deadline)) return false;
        for (int o = first + threadIdx.x; o < nslots; o += NT) st16_agent(g1 + (size_t)b * ws.S + o, v2d{tot[o], tagd});
    }
    return res_gather<NT>(ws, g1, 0, 1, ngroups, ws.S, first, cnt, tagd, tot + first, scr, ctl, deadline);
}

// Block-level sums of the waves' per-lane partial sums through a TRANSPOSE in LDS: lane L of wave w writes its KC column partials
// (and its norm partial) to rows of RES_ROW doubles, four threads per row add 16 values each (stride 4: conflict-free thanks to the
// 4 doubles of padding), one thread per output adds the quarters and the row-waves.  ~0.4 us where KC dependent wave_sum chains took
// 2 us.  T: (NW * KC + NW) * RES_ROW doubles, Q: (NW * KC + NW) * 4 doubles.  Fixed order throughout.
//   vals[jj]: this lane's partial of column c0 + jj, `part` of ED (complex: one call per part) ; nrm: only when part == 0
template <int KC, int NW, int ED>
__device__ __forceinline__ void block_sums(const double (&vals)[KC], double nrm, bool with_cols, int part, int k, int WC, int kcw, double *T,
                                           double *Q, double *mine) {
    constexpr int NT = NW * 64;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int WR = NW / WC;
    if (with_cols) {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) T[(wave * KC + jj) * RES_ROW + lane] = vals[jj];
    }
    if (part == 0) T[(NW * KC + wave) * RES_ROW + lane] = nrm;
    __syncthreads();
    const int row0 = with_cols ? 0 : NW * KC, row1 = part == 0 ? NW * KC + NW : NW * KC;
    for (int idx = row0 * 4 + threadIdx.x; idx < row1 * 4; idx += NT) {
        const double *p = T + (idx >> 2) * RES_ROW + (idx & 3);
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += p[4 * i];
        Q[idx] = s;
    }
    __syncthreads();
    auto rowsum = [&](int row) { return (Q[row * 4] + Q[row * 4 + 1]) + (Q[row * 4 + 2] + Q[row * 4 + 3]); };
    if (with_cols) {
        for (int j = threadIdx.x; j < k; j += NT) {
            const int cg = j / kcw, jj = j - cg * kcw;
            double s = 0.0;
            for (int w = 0; w < WR; ++w) s += rowsum((w * WC + cg) * KC + jj);
            mine[j * ED + part] = s;
        }
    }
    if (part == 0 && threadIdx.x == NT - 1) {
        double s = 0.0;
        for (int w = 0; w < WR; ++w) s += rowsum(NW * KC + w * WC);          // wave-column 0 of every row-wave carries the norm
        mine[k * ED] = s;
        if (ED == 2) mine[k * ED + 1] = 0.0;
    }
    __syncthreads();
}

// acc (per column: real = the lane's two rows, complex = (re, im)) + norm partial -> mine[0, (k + 1) * ED)
template <bool CPLX, int KC, int NW>
__device__ __forceinline__ void block_sums_acc(const v2d (&acc)[KC], double nrm, bool with_cols, int k, int WC, int kcw, double *T, double *Q,
                                               double *mine) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    double v[KC];
    if constexpr (CPLX) {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) v[jj] = acc[jj].x;
        block_sums<KC, NW, ED>(v, nrm, with_cols, 0, k, WC, kcw, T, Q, mine);
        if (with_cols) {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) v[jj] = acc[jj].y;
            block_sums<KC, NW, ED>(v, 0.0, true, 1, k, WC, kcw, T, Q, mine);
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) v[jj] = acc[jj].x + acc[jj].y;
        block_sums<KC, NW, ED>(v, nrm, with_cols, 0, k, WC, kcw, T, Q, mine);
    }
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

// u = sum_jj xv[jj] * h[c0 + jj]: the coefficients are read from LDS at every use (all lanes the same address: a broadcast).  Held in
// registers across the tiles they would cost 2-4 VGPRs per column (they arrive from LDS, not through scalar loads); the index is
// laundered through an empty asm so that the compiler cannot hoist the reads.  Slots beyond nc read column 0 against zeros of X.
template <bool CPLX, int KC>
__device__ __forceinline__ v2d res_project(const v2d (&xv)[KC], const double *h, int c0, int nc) {
    v2d u = v2d{0.0, 0.0};
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) {
        int cj = jj < nc ? c0 + jj : 0;
        asm volatile("" : "+v"(cj));
        if constexpr (CPLX) u += cmul(xv[jj], *reinterpret_cast<const v2d *>(h + 2 * cj));
        else u += xv[jj] * h[cj];
    }
    return u;
}

// the wave-columns' shares of X h for one tile meet in LDS (double buffered: one barrier per tile); every wave of a row-wave ends with
// the same bits
template <int NW>
__device__ __forceinline__ v2d res_exchange(v2d u, v2d *u_lds, int &buf, int WC, int wr) {
    if (WC == 1) return u;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    v2d *ub = u_lds + buf * (NW * 64);
    ub[wave * 64 + lane] = u;
    __syncthreads();
    v2d s = v2d{0.0, 0.0};
    for (int w = 0; w < WC; ++w) s += ub[(wr * WC + w) * 64 + lane];
    buf ^= 1;
    return s;
}

constexpr int res_onchip_tiles(int KC) { return KC >= 16 ? 2 : (KC >= 8 ? 5 : 9); }   // tiles a block of dgs_onchip keeps in registers

struct ResGeom {
    int lane, wave, wc, wr, WR, c0, nc;
    int64_t tile_rows, t0, t1, roff, colstride;
};
template <bool CPLX, int NW>
__device__ __forceinline__ ResGeom res_geom(int k, int64_t n, int64_t ldx, int WC, int kcw) {
    constexpr int ROWS = K<CPLX>::ROWS;
    ResGeom q;
    q.lane = threadIdx.x & 63;
    q.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    q.wc = q.wave % WC;
    q.wr = q.wave / WC;
    q.WR = NW / WC;
    q.c0 = q.wc * kcw;
    int nc = k - q.c0;
    nc = nc > kcw ? kcw : nc;
    q.nc = nc < 0 ? 0 : nc;
    q.tile_rows = (int64_t)q.WR * 64 * ROWS;
    const int64_t ntiles = (n + q.tile_rows - 1) / q.tile_rows;
    q.t0 = ntiles * blockIdx.x / gridDim.x;
    q.t1 = ntiles * (blockIdx.x + 1) / gridDim.x;
    q.roff = (int64_t)q.wr * 64 * ROWS + (int64_t)q.lane * ROWS;
    q.colstride = ldx * K<CPLX>::ELEM_DOUBLES;
    return q;
}

// results and breakdown flag, shared by the two kernels
template <int NTHR>
__device__ __forceinline__ void res_publish(const double *h1, const double *h2, double nrm2, int k, int ED, double *out, int rs, double tol_break,
                                            int *stop_out, int step) {
    if (blockIdx.x != 0) return;
    const int nslots = (k + 1) * ED;
    double *r0 = out, *r1 = out + rs, *r2 = out + 2 * (int64_t)rs;
    for (int o = threadIdx.x; o < nslots; o += NTHR) { r0[o] = h1[o]; r1[o] = h2[o]; }
    if (threadIdx.x == 0) {
        r2[k * ED] = nrm2;
        r2[k * ED + 1] = 0.0;
        if (stop_out && !(sqrt(fabs(nrm2)) >= tol_break)) *stop_out = step;
    }
}
// flags: bit 0 = normalise y'' (skipped below tol_scale) ; bit 1 (dgs_resident) = phase 2 walks the tiles backwards
// out: three result sections of `rs` doubles (h1 | ||y||^2 ; h2 | ||y'||^2 ; slot k = ||y''||^2, slot k*ED+1 of the THIRD section =
// launch status: 0 done, 1 given up before anything was written to y, 2 failed after y'' was stored).
#define LK_RES_ARGS                                                                                                                       \
    const double *__restrict__ X, int64_t ldx, int k, double *__restrict__ y, int64_t n, ResidentWs ws, double *__restrict__ out, int rs, \
        int WC, int kcw, int flags, double tol_scale, double tol_break, int *__restrict__ stop_out, long long spin_ticks, Guard guard

// ---- the panel in registers ------------------------------------------------------------------------------------------------------
// RT tiles of KC columns per wave -- 2 x 16, 5 x 8 or 9 x 4: up to 40 16-byte values = 160 VGPRs per lane -- hold the block's part of X
// for the whole step: 320 KB per CU, 80 MB on the chip.
template <bool CPLX, int KC, int NW>
__global__ __launch_bounds__(NW * 64) void dgs_onchip(LK_RES_ARGS) {
    if (stopped(guard)) return;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int NTHR = NW * 64;
    constexpr int RT = res_onchip_tiles(KC);
    constexpr int CAPS = (KC * NW + 1) * ED + 1;
    __shared__ v2d u_lds[2 * NW * 64];
    __shared__ double T[(NW * KC + NW) * RES_ROW], Q[(NW * KC + NW) * 4];
    __shared__ __attribute__((aligned(16))) double mine[CAPS], h1[CAPS], h2[CAPS], h3[CAPS];
    __shared__ int ctl[2];
    const long long t_entry = wall_clock64();
    const ResGeom q = res_geom<CPLX, NW>(k, n, ldx, WC, kcw);
    const int nb = (int)(q.t1 - q.t0);                     // <= RT (the launcher's grid guarantees it)
    const double *Xw = X + (int64_t)q.c0 * q.colstride;

    // every load of the step is issued before anything else
    v2d xk[RT][KC], yk[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        if (i < nb) {
            const int64_t t = q.t0 + i, r = t * q.tile_rows + q.roff;
            const bool full = (t + 1) * q.tile_rows <= n;
            yk[i] = load_y<CPLX>(y, r, n, full);
            res_load_cols<CPLX, KC, false>(Xw, q.colstride, r, n, full, q.nc, xk[i]);
        } else {
            yk[i] = v2d{0.0, 0.0};
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) xk[i][jj] = v2d{0.0, 0.0};
        }
    }
    if (threadIdx.x == 0) ctl[0] = ld_agent(res_ctr(ws, RES_ABORT)) != 0u;
    __syncthreads();
    if (ctl[0]) return;                                   // a block that became resident after the launch was given up
    const long long start = wall_clock64();
    const long long deadline = start + spin_ticks, deadline_late = deadline + 1000000000ll;
    double *r2 = out + 2 * (int64_t)rs;
    auto give_up = [&](double status) {
        if (threadIdx.x == 0) {
            r2[k * ED + 1] = status;
            if (stop_out) *stop_out = guard.step;
        }
    };
    auto stamp = [&](int i) { if (blockIdx.x == 0 && threadIdx.x == 0) ws.tim[i] = wall_clock64(); };
    if (blockIdx.x == 0 && threadIdx.x == 0) ws.tim[0] = t_entry;   // (taken at kernel entry: the abort-word load above returns behind every tile load,
                                                                    //  a stamp here would hide the whole panel read from phase 1)

    // accumulators: complex (re, im); real ONE double per column, the lane's two rows through the same chain (32 VGPRs less)
    v2d acc[KC];
    double nrm = 0.0;
    auto dots = [&]() {
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) acc[jj] = v2d{0.0, 0.0};
        nrm = 0.0;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if (i < nb) {
#pragma unroll
                for (int jj = 0; jj < KC; ++jj) {
                    if constexpr (CPLX) acc[jj] += cmulconj(xk[i][jj], yk[i]);
                    else acc[jj].x = fma(xk[i][jj].y, yk[i].y, fma(xk[i][jj].x, yk[i].x, acc[jj].x));
                }
                nrm += yk[i].x * yk[i].x + yk[i].y * yk[i].y;
            }
        }
    };
    int buf = 0;
    auto update = [&](const double *h) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if (i < nb) {                                  // (block-uniform: the barrier inside res_exchange is safe)
                const v2d u = res_project<CPLX, KC>(xk[i], h, q.c0, q.nc);
                yk[i] -= res_exchange<NW>(u, u_lds, buf, WC, q.wr);
            }
        }
    };

    dots();                                                                                   // phase 1
    block_sums_acc<CPLX, KC, NW>(acc, nrm, true, k, WC, kcw, T, Q, mine);
    stamp(1);
    if (spin_ticks == 0) {            // "resident_spin_ms" = 0: give up at the first wait without waiting (how the tests reach the fallback)
        if (threadIdx.x == 0) __hip_atomic_store(res_ctr(ws, RES_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        give_up(1.0);
        return;
    }
    if (!grid_sum<NTHR>(ws, 0, 0, (k + 1) * ED, mine, h1, ctl, T, deadline)) { give_up(1.0); return; }
    stamp(2);
    update(h1);                                                                               // phase 2
    dots();
    block_sums_acc<CPLX, KC, NW>(acc, nrm, true, k, WC, kcw, T, Q, mine);
    stamp(3);
    if (!grid_sum<NTHR>(ws, 1, 0, (k + 1) * ED, mine, h2, ctl, T, deadline_late)) { give_up(1.0); return; }
    stamp(4);
    update(h2);                                                                               // phase 3
    nrm = 0.0;
#pragma unroll
    for (int i = 0; i < RT; ++i)
        if (i < nb) nrm += yk[i].x * yk[i].x + yk[i].y * yk[i].y;
    block_sums_acc<CPLX, KC, NW>(acc, nrm, false, k, WC, kcw, T, Q, mine);
    stamp(5);
    if (!grid_sum<NTHR>(ws, 2, k * ED, (k + 1) * ED, mine, h3, ctl, T, deadline_late)) { give_up(1.0); return; }   // (y not stored yet: still a clean give-up)
    stamp(6);
    const double nr = sqrt(fabs(h3[k * ED]));
    res_publish<NTHR>(h1, h2, h3[k * ED], k, ED, out, rs, tol_break, stop_out, guard.step);
    const double scale = ((flags & 1) && nr >= tol_scale) ? 1.0 / nr : 1.0;                   // phase 4: y'' leaves the chip ONCE, scaled
    if (q.wc == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if (i < nb) {
                const int64_t t = q.t0 + i, r = t * q.tile_rows + q.roff;
                store_rows<CPLX>(y, r, n, (t + 1) * q.tile_rows <= n, yk[i] * scale, 0);
            }
        }
    }
    stamp(7);
}

// ---- the panel in the caches -------------------------------------------------------------------------------------------------------
// One phase over the block's tiles [t0, t1): panel_sweep's tile body (SC = 1, G = 1).
//   MODE 1: dot   MODE 2: update (kept in registers) + dot   MODE 4: two-coefficient update, stored
template <bool CPLX, int KC, int NW, int MODE>
__device__ __forceinline__ void res_phase(const double *__restrict__ X, double *__restrict__ y, int64_t n, int k, const ResGeom &q,
                                          const double *h1, const double *h2, int WC, int kcw, bool reverse, v2d *u_lds, double *T, double *Q,
                                          double *mine) {
    constexpr bool UPDATE = MODE != 1, DOT = MODE <= 2, TWO = MODE == 4;
    v2d acc[KC];
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) acc[jj] = v2d{0.0, 0.0};
    double nrm = 0.0;
    const double *Xw = X + (int64_t)q.c0 * q.colstride;
    int buf = 0;
    for (int64_t i = q.t0; i < q.t1; ++i) {
        const int64_t t = reverse ? q.t1 - 1 - (i - q.t0) : i;
        const int64_t r = t * q.tile_rows + q.roff;
        const bool full = (t + 1) * q.tile_rows <= n;
        v2d xv[KC];
        v2d yv = load_y<CPLX>(y, r, n, full);
        res_load_cols<CPLX, KC, false>(Xw, q.colstride, r, n, full, q.nc, xv);
        if constexpr (UPDATE) {
            const v2d u = res_project<CPLX, KC>(xv, h1, q.c0, q.nc);
            yv -= res_exchange<NW>(u, u_lds, buf, WC, q.wr);
            if constexpr (TWO) {
                const v2d u2 = res_project<CPLX, KC>(xv, h2, q.c0, q.nc);
                yv -= res_exchange<NW>(u2, u_lds, buf, WC, q.wr);
                if (q.wc == 0) store_rows<CPLX>(y, r, n, full, yv, 0);
            }
        }
        if constexpr (DOT) {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                if constexpr (CPLX) acc[jj] += cmulconj(xv[jj], yv);
                else acc[jj] += xv[jj] * yv;
            }
        }
        nrm += yv.x * yv.x + yv.y * yv.y;
    }
    block_sums_acc<CPLX, KC, NW>(acc, nrm, DOT, k, WC, kcw, T, Q, mine);
}

template <bool CPLX, int KC, int NW>
__global__ __launch_bounds__(NW * 64) void dgs_resident(LK_RES_ARGS) {
    if (stopped(guard)) return;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr int NTHR = NW * 64;
    constexpr int CAPS = (KC * NW + 1) * ED + 1;
    __shared__ v2d u_lds[2 * NW * 64];
    __shared__ double T[(NW * KC + NW) * RES_ROW], Q[(NW * KC + NW) * 4];
    __shared__ __attribute__((aligned(16))) double mine[CAPS], h1[CAPS], h2[CAPS], h3[CAPS];
    __shared__ int ctl[2];
    if (threadIdx.x == 0) ctl[0] = ld_agent(res_ctr(ws, RES_ABORT)) != 0u;
    __syncthreads();
    if (ctl[0]) return;                                   // a block that became resident after the launch was given up
    // the FIRST wait is the one that can starve (a block not yet resident); past it every block is on the chip.  The later waits
    // still carry a (generous) bound so that a defect can never hang the device; one firing there is reported as a failure.
    const long long start = wall_clock64();
    const long long deadline = start + spin_ticks, deadline_late = deadline + 1000000000ll;
    const ResGeom q = res_geom<CPLX, NW>(k, n, ldx, WC, kcw);
    const int nslots = (k + 1) * ED;
    double *r2 = out + 2 * (int64_t)rs;
    auto give_up = [&](double status) {
        if (threadIdx.x == 0) {
            r2[k * ED + 1] = status;
            if (stop_out) *stop_out = guard.step;
        }
    };
    auto stamp = [&](int i) { if (blockIdx.x == 0 && threadIdx.x == 0) ws.tim[i] = wall_clock64(); };
    stamp(0);
    res_phase<CPLX, KC, NW, 1>(X, y, n, k, q, nullptr, nullptr, WC, kcw, false, u_lds, T, Q, mine);
    stamp(1);
    if (spin_ticks == 0) {            // "resident_spin_ms" = 0: give up at the first wait without waiting (how the tests reach the fallback)
        if (threadIdx.x == 0) __hip_atomic_store(res_ctr(ws, RES_ABORT), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        give_up(1.0);
        return;
    }
    if (!grid_sum<NTHR>(ws, 0, 0, nslots, mine, h1, ctl, T, deadline)) { give_up(1.0); return; }
    stamp(2);
    res_phase<CPLX, KC, NW, 2>(X, y, n, k, q, h1, nullptr, WC, kcw, (flags & 2) != 0, u_lds, T, Q, mine);
    stamp(3);
    if (!grid_sum<NTHR>(ws, 1, 0, nslots, mine, h2, ctl, T, deadline_late)) { give_up(1.0); return; }
    stamp(4);
    res_phase<CPLX, KC, NW, 4>(X, y, n, k, q, h1, h2, WC, kcw, false, u_lds, T, Q, mine);
    stamp(5);
    if (!grid_sum<NTHR>(ws, 2, k * ED, nslots, mine, h3, ctl, T, deadline_late)) { give_up(2.0); return; }
    stamp(6);
    const double nr = sqrt(fabs(h3[k * ED]));
    res_publish<NTHR>(h1, h2, h3[k * ED], k, ED, out, rs, tol_break, stop_out, guard.step);
    if ((flags & 1) && nr >= tol_scale) {
        // the lanes that stored y'' scale it: same wave, same lane, same address as the store (ordered by the hardware)
        const double inv = 1.0 / nr;
        if (q.wc == 0) {
            for (int64_t t = q.t0; t < q.t1; ++t) {
                const int64_t r = t * q.tile_rows + q.roff;
                const bool full = (t + 1) * q.tile_rows <= n;
                v2d yv = load_y_l2<CPLX>(y, r, n, full);
                yv *= inv;
                store_rows<CPLX>(y, r, n, full, yv, 0);
            }
        }
    }
    stamp(7);
}

}  // namespace lk
