// lk_engine.hip -- host side of the C ABI declared in include/lightkrylov_hip.h.
// HIP only: there is no CPU code path in this library.
#include "lk_internal.h"
#include "lk_kernels.hip.h"
#include "lk_resident.hip.h"
#include <hip/hip_ext.h>
#include <type_traits>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace lk;

namespace {

constexpr double ATOL_DP = 1.0e-15;  // src/Constants.f90:35  atol_dp = 10**(-precision(1.0_dp))
constexpr int KMAX_FUSED = 128;      // columns one fused sweep can hold (KC * NW) without the lane split
constexpr int KMAX_WIDE = 512;       // ... with the lanes of a wave split 4 ways over column groups (panel_sweep's SC)
constexpr int MAX_GRID = 4096;       // upper bound on sweep blocks (partial buffer stride)

thread_local char g_err[512] = "";

#define fail lk_fail_

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (void)hipGetLastError(); /* the runtime's last-error state is sticky: report once */ \
            return fail(LK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                    \
    } while (0)

#define LKCHK(expr)            \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != LK_OK) return rc_; \
    } while (0)

std::mutex g_ctx_mu;
std::set<void *> g_live_ctx;   // lk_basis_destroy may run after lk_finalize (garbage collectors): never touch a dead context

struct ProfRec {
    hipEvent_t e0, e1;
    std::string tag;
    double bytes;
    bool borrowed = false;   // e0/e1 belong to other records (a span built from their events): not returned to the pool twice
};
struct ProfAcc {
    int64_t count = 0;
    double ms = 0.0, bytes = 0.0;
};

}  // namespace

extern "C" int lk_fail_(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct lk_context_s {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    int grid_mult = 2;  // sweep blocks per CU
    int stream_update = 1;     // single-coefficient update sweeps: barrier-free streaming kernel
    int update_grid_mult = 4;
    int gemm_grid_mult = 4;    // panel_gemm blocks per CU
    int gemm_mfma = 1;         // tall-skinny product on the FP64 matrix cores (0: FP64 VALU kernel)
    int gemm_mfma_min = 0;     // ... for at least this many output columns (0: 5 real / 9 complex, from profiles/r03_lincomb_scan.txt);
                               // narrower products stream X through the VALU kernel with 1 / 2 / 4 / 8 accumulators per lane
    int gemm_store_policy = 2; // cache policy of the product's output stores (as store_policy: 2 = sc1 write-through)
    int stream_two = 0;        // sweep 3 with two coefficient sets: barrier-free streaming kernel instead of the LDS/barrier one
    int recompute_update = 1;  // two-pass DGS: sweep 2 does not store y'; sweep 3 re-forms it (3k+4 instead of 3k+5 columns)
    int store_policy = 2;      // cache policy of the sweeps' y store: 0 plain, 1 nt, 2 sc1 (write-through; +2% on sweep 3), 3 sc0 sc1
    int store_split = 0;       // every wave of the column split stores a lane slice instead of the wc == 0 wave
    int gemm_prefetch_y = 1;   // accumulating real MFMA product (<= 32 outputs): load the tile of Y ahead of the k-loop (0: after it)
    int upd_rs = 1;            // fused pass of the block DGS, real kind, 17..32 right-hand sides: 1 = panel_xhy_upd_rs (row-owner waves, LDS-DMA tiles, coefficients in registers), 0 = panel_xhy_upd_mfma
    int gram_rs = 1;           // Gram matrix of 5..128 real columns by panel_gram_rs (rows of the staged tile dealt to the waves, LDS-DMA staging; n = 10^7: k = 8 0.29 -> 0.12 ms,
                               // k = 16 0.35 -> 0.21, k = 48 1.11 -> 0.63, k = 96 2.81 -> 1.80, k = 128 3.45 -> 3.0) and of 5..112 complex columns by panel_gram_rs3m / rs3m4 (n = 5 10^6: k = 16 0.46 -> 0.22, k = 48 1.35 -> 0.84, k = 96 3.33 -> 2.69):
                               // 1 = as many blocks per CU as are resident, n > 1 = n blocks per CU, 0 = panel_xhy_mfma (one tile row per wave) / panel_gram_mfma3m
    int upd_debug = 0;         // -DLK_DIAGNOSTICS builds only (WRONG results, phase timing): panel_xhy_upd_mfma without 1 = the update MFMAs, 2 = the dot MFMAs, 4 = the global loads after the first tile, 8 = the store of Y'
    int xhy_debug = 0;         // -DLK_DIAGNOSTICS builds only (WRONG results, phase timing): 1 = panel_xhy_mfma without its MFMAs, 2 = without the global loads after the first tile
    int gemm_roll = 1;         // MFMA tall-skinny product: rolling prefetch of X (a ring of 4 k-steps refilled as they are consumed, carried across tiles) instead of batches of 4
                               // k-steps (loaded, waited for, multiplied).  1: the real kind with 33..64 outputs per pass, on the STRAIGHT-LINE ring (no branch between loads and MFMAs,
                               // exact vmcnt counts; k = 128, q = 64 at n = 10^7: 3.45-3.73 -> 3.16-3.24 ms); 2: every variant that has a ring (the narrow real ones measure the same
                               // as the batch schedule, the complex doubled-real ones keep the guarded ring, +2-5 % on narrow products); 0: never (profiles/r05_ab_gemm_roll.jsonl)
    int xhy_db = 1;            // panel_xhy_mfma with a double-buffered LDS tile (1: the 128-column variants; 2: the <= 32 right-hand-side ones too; 0: never)
    int gemm_3m = 1;           // complex MFMA kernels (tall-skinny product; X^H Y with <= 32 right-hand sides; Gram) with three real products per complex one (0: four, the doubled real problem)
    int kc32 = -1;             // real DGS update sweeps (2 and 3) of k > kc32 (<= 128) columns on 32-column register tiles (0: never; -1: k > 32 when the GLOBAL problem has >= 2^25 rows)
    int wide_s3 = 1;           // sweep 3 of a lane-split (SC = 2) DGS with both column groups of a wave-column in one wave's registers (G = 2), tiles twice as tall
    int wide_regs = 2;         // wide REGISTER tiles: 1 = 8 waves x 32 / 24 columns for 129..256 real / 129..192 complex basis columns instead of the lane split; 2 = also the lane split on 24-column groups for 257..384 columns; 0 = round 3's shapes
    int cplx_wide = 32;        // complex sweeps with 8 waves x 16 columns per block when k exceeds this (0: never) instead of 16 x 8
    // reduction workspace
    double *partial = nullptr;  // [(KMAX_FUSED+1)*2][MAX_GRID]
    double *red = nullptr;      // device results: 3 sections of (KMAX_FUSED+1)*2 doubles
    double *red_host = nullptr; // pinned mirror
    double *coef = nullptr;     // device coefficients of the lazy path's pending updates (KMAX_WIDE*2 doubles)
    double *scratch = nullptr;  // scratch vector (grown on demand), scratch_n doubles
    int64_t scratch_n = 0;
    double *lz_red = nullptr, *lz_red_host = nullptr;   // lk_lanczos: per step 4 sections (two local passes x {dot, update})
    int lz_cap = 0;
    double *xhy = nullptr;      // panel_xhy_mfma: [2 result sections][norm partials][partials], grown on demand
    int64_t xhy_n = 0;
    // communication
    lk_allreduce_fn allreduce = nullptr;
    void *allreduce_user = nullptr;
    lk_halo_fn halo = nullptr;
    void *halo_user = nullptr;
    lk_allgather_fn allgather = nullptr;
    void *allgather_user = nullptr;
    int nranks = 1, rank = 0;
    int64_t row0 = 0, n_global = -1;  // this rank's row block [row0, row0 + n_local) of n_global rows
    // lazy batching of the PER-OBJECT path (opt-in, tuning key "lazy"): what an unchanged LightKrylov drives
    // through the type-bound procedures -- k consecutive X(i)%dot(y), then k consecutive y%axpby(a_i, X(i), 1).
    int lap5_grid_mult = 8;    // persistent blocks per CU of the stencil operator
    int dot_colwise = 1;       // sweep 1 by panel_dot_cw (one column at a time, y in registers) instead of panel_sweep<DOT>
    int grid_mult_s3 = 0;      // blocks per CU of the two-coefficient update sweep (0: grid_mult)
    int grid_mult_s2 = 1;      // blocks per CU of the update + dot sweep (0: grid_mult).  1 is never slower than 2 and +1-6 % at small n or small k
    int cw_u = 0;              // its 16-byte loads per lane and column: 4, 8, or 0 = by size (8 on 2 blocks per CU for long panels)
    int cw_grid_mult = 3;      // its blocks per CU (A/B at n = 10^8: 3 > 4 > 6)
    int xcd_map = 0;           // A/B: contiguous eighth of the rows per XCD instead of grid-cyclic tiles (null: DESIGN tuning log)
    int prof_ext = 1;          // profiling events of the sweeps attached to the kernel dispatch instead of recorded on the stream
    int xhy_grid_mult = 0;     // blocks per CU of the 32-row-tile variant (0: 3 real / 2 complex)
    int xhy_small = 1;         // its 32-row-tile variant for <= 32 right-hand sides
    int xhy_mfma = 1;          // X^H Y with >= XHY_MIN_P right-hand sides (Gram, innerprod_matrix, block DGS) on the FP64 matrix cores
    int block_fused = 1;       // block DGS: fused update+dot / two-coefficient sweeps (3 passes per group) instead of 4
    int csr_stream = 1;        // CSR product through LDS for matrices with short rows (mean <= 32 entries); 0: lanes-per-row kernel
    int csr_lanes = 0;         // 0: lanes per row of the CSR product chosen from the mean row length; 2..64 forces it
    int blas1_grid_mult = 2;   // blocks of 256 threads per CU for the one-to-three-stream kernels
    // single-launch Gram-Schmidt step for cache-resident panels (lk_resident.hip.h)
    int resident = 1;          // 0: never; 1: when the panel X(:, :k) | y fits `resident_max_mb` (one rank only: the phases meet inside the launch)
    int resident_max_mb = 320; // ... MB of panel the single launch takes (measured crossover with the three sweeps: profiles/r06_resident_phases.jsonl)
    int resident_onchip = 1;   // panels that fit the register files (64 MB on the chip) stay there for the whole step: X is read once
    int resident_rev = 1;      // phase 2 walks a block's tiles backwards (starts on what phase 1 read last)
    int resident_spin_ms = 50;     // bound on the first grid-wide wait (it normally ends within microseconds); beyond it the launch gives up, the
                               // three-sweep schedule runs that step
    bool resident_off = false; // a launch gave up (the device is shared with another persistent kernel): the single launch pauses ...
    int64_t resident_pause = 16;       // ... for this many Gram-Schmidt steps (doubling with every give-up, up to 2^20), then is tried again
    int64_t resident_fallback_steps = 0, resident_retry_at = 0;   // steps run on the three sweeps while paused; the count at which to re-arm
    unsigned *res_cnt = nullptr;
    long long *res_tim = nullptr;
    double *blk_red = nullptr, *blk_red_host = nullptr;   // coefficient sections of the block Gram-Schmidt (lk_dgs_block, lk_arnoldi_block): device + pinned
    int64_t blk_cap = 0;
    void *res_gran = nullptr;              // {value, tag} granules of the grid sums
    unsigned long long res_epoch = 0;      // launches so far (the tag of a launch's granules)
    int64_t resident_stats[3] = {0, 0, 0};   // single launches enqueued, launches that gave up, launches that kept the panel in registers
    int lazy = 0;
    struct {
        bool valid = false;
        const double *xbase = nullptr;   // panel the memoised columns live in
        const double *y = nullptr;       // the vector they were dotted with
        int j0 = 0, cnt = 0;
        std::vector<double> vals;
    } memo;
    // Gram memo: the reference's gram_matrix asks X(i)%dot(X(j)), j = i..k, i = 1..k (AbstractVectors.fypp:651-656): self fixed,
    // vec running, so the batched-dot memo above (vec fixed) never applies and is_orthonormal would cost k(k+1)/2 dot kernels.
    // The second call of such a run computes X^H X of the written columns once on the matrix cores; the rest are hits.
    struct {
        bool valid = false;
        const double *xbase = nullptr;
        int cnt = 0;
        std::vector<double> G;           // [j][i][ED] = conj(X_i) . X_j
        const double *last_x = nullptr;  // the previous in-panel dot call (pattern detection)
        int last_jx = -1, last_jy = -1;
    } gmemo;
    void forget_memos() { memo.valid = false; nmemo.valid = false; gmemo.valid = false; gmemo.last_x = nullptr; }
    // queue: pending  T <- [zeroed ? 0 : T] + sum_i a_i X(:, j0+i)  (linear_combination's loop).  `zeroed`: the queue
    // began with T%zero() (which was itself deferred), so T is DEFINED by the queue and can stay VIRTUAL -- never
    // written -- as long as nobody reads it: linear_combination's temporary is consumed by one y%sub(proj) and dies.
    // By == nullptr: T's storage is already gone (its 1-column panel was destroyed); the coefficients live on for `sub`.
    struct {
        bool active = false, zeroed = false;
        lk_basis_t Bx = nullptr, By = nullptr;
        int j0 = 0, cnt = 0, jy = 0;
        std::vector<double> coef;        // a_i, ED doubles each
    } queue;
    // sub: pending  y <- y + s * T  with T the (zeroed) queue's still virtual target, i.e. y += s X a.  Applied as ONE
    // fused sweep together with the dots and the norm the NEXT Gram-Schmidt pass asks for (y%norm(), X(i)%dot(y)).
    struct {
        bool active = false;
        lk_basis_t By = nullptr;
        int jy = 0;
        double s[2] = {0.0, 0.0};
    } sub;
    struct { bool valid = false; const double *y = nullptr; double nrm2 = 0.0; } nmemo;   // ||y||^2 from the last fused sweep
    // first pass of a Gram-Schmidt step, anticipated.  orthogonalize_against_basis opens with y%norm() (gram_schmidt.fypp:126) and
    // then asks X(1..k)%dot(y): a norm kernel + a host synchronisation, then the batched dot sweep + another one -- although the
    // sweep computes ||y||^2 on the side.  Once that pair has been SEEN for y = column j of a panel (norm of column j, then the
    // dots of columns [0, j) against it), the norm of column j + 1 of the same panel -- the next Arnoldi / Lanczos step -- runs
    // the sweep over [0, j + 1) at once and serves norm and dots from it: one kernel and one synchronisation less per step.
    // A prediction nobody used (no dot of the batch was asked for before the next one) disarms it: at most one wasted sweep.
    struct {
        const double *xbase = nullptr;   // panel the pattern was seen on
        int jy = -1, j0 = 0;             // ... for this column, against the columns [j0, jy)
        bool armed = false, unused = false;
        const double *last_norm_y = nullptr;   // the vector whose norm was the previous lazy-mode call (plain kernel)
    } spec;
    int lazy_speculate = 1;              // tuning key: 0 switches the anticipation off
    int64_t fusion_stats[4] = {0, 0, 0, 0};  // fused update+dot sweeps, plain deferred updates, virtual temporaries dropped, materialised
    int64_t spec_stats[2] = {0, 0};          // anticipated first-pass sweeps, of which unused
    double *coef_host = nullptr;         // pinned staging for queued coefficients
    hipEvent_t coef_ev = nullptr;        // completion of the last staging copy
    int64_t lazy_stats[4] = {0, 0, 0, 0};  // dot memo hits, batched dot sweeps, queued axpbys, queue flushes
    // asynchronous Arnoldi pipeline (lk_arnoldi): per-step result slots + device-side breakdown flag
    int *stop_dev = nullptr;               // device int: 0, or the step that asked every later step to stop
    int *stop_host = nullptr;              // pinned mirror
    int *seg_stop_host = nullptr;          // pinned: the stop flag as it stood after each segment of a segmented batch (lk_arnoldi_segments)
    std::vector<hipEvent_t> seg_events;    // ... and the event recorded behind each segment's copies
    int seg_cap = 0;
    bool guard_on = false;                 // launches carry the guard only inside an asynchronous batch
    int guard_step = 0;
    double *step_red = nullptr;            // device: nsteps x RED_SECTIONS x RED_SECTION doubles
    double *step_red_host = nullptr;       // pinned mirror
    int64_t step_red_cap = 0;              // doubles the two buffers hold
    int async_arnoldi = 1;                 // tuning key: 0 = one host round trip per step (the round-1 schedule)
    Guard guard() const { return Guard{guard_on ? stop_dev : nullptr, guard_step}; }
    // column pool (lk_pool_*): slabs handed out to per-object hosts
    struct PoolSlab {
        lk_basis_t B = nullptr;
        int used = 0;                          // columns carved so far
        std::vector<uint64_t> owner;           // owner tag per column (0 = free)
        std::vector<uint64_t> gen;             // generation per column: a fresh value of pool_epoch every time lk_pool_acquire hands the column out
    };
    // Generations are drawn from ONE counter per context that only ever grows -- never per slab, never reset by
    // lk_pool_release_all: a slab allocated later can land on a freed slab's heap address, and with per-slab counters its column
    // j would be handed out with generation 1 again; a stale handle from before the release (same address, column, generation)
    // would then pass as the new occupant's.
    uint64_t pool_epoch = 0;
    std::vector<PoolSlab> pool;
    std::map<uint64_t, std::pair<int, int>> pool_by_tag;   // owner tag -> (slab index, column)
    std::set<std::pair<int, int>> pool_free;                // released columns, lowest first
    int pool_slab_cols = 160;
    int64_t pool_stats[2] = {0, 0};            // columns ever carved, acquisitions served by re-use
    // profiling
    bool prof_sweeps_only = false;   // inside the asynchronous Arnoldi batch only the sweeps carry events (each costs ~4 us of queue time)
    hipEvent_t span_first = nullptr, span_last = nullptr;   // first start / last stop event recorded since span_first was cleared
    bool prof = false;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    std::map<std::string, ProfAcc> prof_acc;
};

struct lk_basis_s {
    lk_context_t ctx;
    int dtype;
    int64_t n, ld;
    int ncols;
    double *data;
    bool own;
    int hwm = 0;   // columns [0, hwm) have been written through the ABI (lazy dot batches never sweep beyond it)
    void touch(int j, int cnt = 1) { if (j + cnt > hwm) hwm = j + cnt; }
    int ed() const { return dtype == LK_C128 ? 2 : 1; }
    double *col(int j) const { return data + (int64_t)j * ld * ed(); }
};

enum OpKind { OP_DIAG, OP_DIAG_LIN, OP_DENSE, OP_LAP5, OP_GL, OP_CSR };
struct lk_linop_s {
    lk_context_t ctx;
    OpKind kind;
    int dtype;
    int64_t n;
    double *dev = nullptr;  // diag values / dense matrix
    bool own_dev = true;    // dense: the matrix memory belongs to the operator (false: wrapped caller memory)
    int64_t lda = 0;
    // row-sharded dense / CSR: this rank holds rows [rstart[rank], rstart[rank+1]) of an ncols_g x ncols_g operator
    int64_t ncols_g = 0;                  // global size (= n on a single rank)
    std::vector<int64_t> gcounts, gdispls;   // all-gather layout in DOUBLES (per rank)
    double *xfull = nullptr;              // the gathered input vector / the full-length adjoint product (ncols_g elements)
    double *gpart = nullptr;              // k_gemv_n's per-chunk partial sums
    int gchunks = 1;
    // row-sharded CSR, COMPRESSED exchange: only the entries of x that some other rank's rows reference travel.  cx_send_idx =
    // local indices of this rank's entries anybody needs (sorted); every rank's packed entries are all-gathered into cx_xrem
    // (rank r's at cx_displs[r]); the column indices of csr[0] address [own rows | cx_xrem].
    bool cx = false;
    int32_t *cx_send_idx = nullptr;
    int64_t cx_nsend = 0, cx_total = 0;
    double *cx_sendbuf = nullptr, *cx_xrem = nullptr;
    std::vector<int64_t> cx_counts, cx_displs;   // in DOUBLES, per rank
    int64_t row0 = 0;
    double d0 = 0, dstep = 0;
    int64_t N = 0;
    // Ginzburg-Landau stepper
    double gl[8] = {0};   // dx, halfL, nu_re, nu_im, ga_re, ga_im, mu_c, mu2
    double tau = 0;
    int nsub = 1;
    double *wk = nullptr; // 3 work vectors (k_a, k_b, u_sub)
    // row-sharded stencil operators
    int64_t NJ = 0;            // lap5: grid lines held by this rank
    int64_t n_global = 0;      // GL: global rows
    bool has_lo = false, has_hi = false;   // a neighbouring rank below / above this block
    double *halo = nullptr;    // lap5: 2 N doubles (line from rank-1 | line from rank+1); GL: 4 doubles
    double *edges = nullptr;   // GL: this rank's two edge values (send buffer), 4 doubles
    // CSR: A (for 'N') and its conjugate transpose (for 'H'), both row-compressed; W = lanes per row
    struct Csr {
        int64_t *rowptr = nullptr; int32_t *colind = nullptr; double *vals = nullptr; int W = 8;
        int64_t *rowblocks = nullptr; int64_t nblocks = 0, nnz_hint = 0;   // CSR-stream partition (short rows), see k_csr_stream
    } csr[2];
};

namespace {

constexpr int RED_SECTION = (KMAX_FUSED + 1) * 2;  // doubles per result section
constexpr int RED_SECTION_WIDE = (KMAX_WIDE + 1) * 2;   // ... of a sweep over a wide basis (129..512 columns)
inline int red_stride(int k) { return k <= KMAX_FUSED ? RED_SECTION : RED_SECTION_WIDE; }
constexpr int RED_SECTIONS = 3;                     // h1 | h2 | ||y''||^2 of one vector DGS (also the per-step slot of lk_arnoldi)
constexpr int RED_MULTI = 4;                        // sections one multi-RHS dot pass fills: a flat [4][k+1] buffer
constexpr int RED_TOTAL = 2 * RED_MULTI + 1;        // sections of c->red: two such passes (block DGS keeps both on the device); the
                                                    // wide single-vector DGS uses up to 4 + 4 panels of coefficients + one norm section
constexpr int PARTIAL_SECTIONS = 4;                 // per-block partials: up to 4 y-columns per multi-RHS pass

// ---- profiling helpers ----------------------------------------------------------------
struct ProfScope {
    lk_context_t c;
    bool on;
    ProfRec rec;
    bool ext = false;   // the launch itself carries the two events (hipExtLaunchKernelGGL): nothing is recorded on the stream
    ProfScope(lk_context_t ctx, const char *tag, double bytes, bool ext_launch = false, bool enable = true) : c(ctx), on(ctx->prof && enable), ext(ext_launch) {
        if (on && c->prof_sweeps_only && strncmp(tag, "dgs_sweep", 9) != 0 && strcmp(tag, "matvec") != 0 && strncmp(tag, "comm_", 5) != 0) on = false;
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        rec.e0 = get();
        rec.e1 = get();
        rec.tag = tag;
        rec.bytes = bytes;
        if (!ext) (void)hipEventRecord(rec.e0, c->stream);
    }
    void end() {
        if (!on) return;
        on = false;
        if (!ext) (void)hipEventRecord(rec.e1, c->stream);
        if (!c->span_first) c->span_first = rec.e0;
        c->span_last = rec.e1;
        c->prof_pending.push_back(rec);
    }
    ~ProfScope() { end(); }
};

void prof_collect(lk_context_t c) {
    if (c->prof_sweeps_only) return;      // an asynchronous batch is being enqueued: its span records still reference live events
    for (auto &r : c->prof_pending) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        auto &a = c->prof_acc[r.tag];
        a.count += 1;
        a.ms += ms;
        a.bytes += r.bytes;
        if (!r.borrowed) {
            c->ev_pool.push_back(r.e0);
            c->ev_pool.push_back(r.e1);
        }
    }
    c->prof_pending.clear();
}

// Every ABI entry that touches the device runs on the CONTEXT's device, whatever the caller's current device is
// (a process may hold contexts on several GPUs, or torch may have switched the current device).
struct DevGuard {
    int prev = -1;
    bool switched = false;
    explicit DevGuard(lk_context_t c) {
        if (!c) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DevGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// non-temporal accesses for vectors that cannot stay in the 32 MB of L2 anyway (the next kernel re-reads them from HBM)
inline int blas1_nt(lk_basis_t B) { return (int64_t)B->n * B->ed() * 8 >= (int64_t)32 << 20; }
inline int blas1_grid(lk_context_t c, int64_t nvec) {
    int64_t g = (nvec + 255) / 256;
    int64_t cap = (int64_t)c->num_cu * c->blas1_grid_mult;   // 2 blocks per CU: lk_kernels.hip.h, BLAS-1 header
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int check_vec(lk_basis_t B, int j, const char *what) {
    if (!B) return fail(LK_ERR_INVALID, "%s: null basis", what);
    if (j < 0 || j >= B->ncols) return fail(LK_ERR_INVALID, "%s: column %d out of range [0,%d)", what, j, B->ncols);
    return LK_OK;
}
int check_pair(lk_basis_t A, lk_basis_t B, const char *what) {
    if (A->ctx != B->ctx) return fail(LK_ERR_INVALID, "%s: bases belong to different contexts", what);
    if (A->dtype != B->dtype) return fail(LK_ERR_INVALID, "%s: dtype mismatch", what);
    if (A->n != B->n) return fail(LK_ERR_INVALID, "%s: Inconsistent size between the two vectors (%lld vs %lld)", what,
                                  (long long)A->n, (long long)B->n);
    return LK_OK;
}

int allreduce(lk_context_t c, double *dev, int64_t count) {
    if (c->nranks > 1 && !c->allreduce) return fail(LK_ERR_COMM, "nranks=%d but no all-reduce installed", c->nranks);
    if (c->allreduce) {
        // "comm_allreduce": stream markers either side of the collective -- on the native route the time the ncclAllReduce kernel
        // occupies the engine's stream (launch + ring latency + the wait for the slowest peer), on a host route the whole round trip
        ProfScope ps(c, "comm_allreduce", (double)count * 8.0);
        int rc = c->allreduce(c->allreduce_user, dev, count, (void *)c->stream);
        if (rc != 0) return fail(LK_ERR_COMM, "all-reduce callback returned %d", rc);
    }
    return LK_OK;
}

// ---- sweep launcher ---------------------------------------------------------------------
struct SweepCfg { int WC, kcw, grid; int64_t ntiles; };

template <bool CPLX, int KC, int NW, int SC = 1>
SweepCfg sweep_cfg(lk_context_t c, int k, int64_t n, int mult = 0) {
    SweepCfg s;
    int cg = (k + KC - 1) / KC;              // column groups needed (one per wave, or per lane group of a wave when SC > 1)
    int wc = (cg + SC - 1) / SC;             // waves needed across columns
    if (wc < 1) wc = 1;
    int WC = 1;
    while (WC < wc) WC <<= 1;    // power of two so it divides NW
    if (WC > NW) WC = NW;
    s.WC = WC;
    s.kcw = (k + WC * SC - 1) / (WC * SC);
    if (s.kcw < 1) s.kcw = 1;
    const int64_t tile_rows = (int64_t)(NW / WC) * (64 / SC) * K<CPLX>::ROWS;
    int64_t ntiles = (n + tile_rows - 1) / tile_rows;
    int64_t g = (int64_t)c->num_cu * (mult > 0 ? mult : c->grid_mult);
    if (g > ntiles) g = ntiles;
    if (g > MAX_GRID) g = MAX_GRID;
    if (g < 1) g = 1;
    s.grid = (int)g;
    s.ntiles = ntiles;
    return s;
}

// One sweep over columns [0,k) of X (k <= KMAX_FUSED) + finish into `out` (k+1 slots of ED doubles).
//   MODE 1: h = X^H y                       (DOT)
//   MODE 2: y' = y - X hin ; h = X^H y'     (UPDATE+DOT), y' written only if `store`
//   MODE 3: y' = y - X hin                  (UPDATE)
//   MODE 4: y'' = (y - X hin) - X hin2      (UPDATE, two coefficient sets; pairs with MODE 2, store = 0)
// out == nullptr (update-only modes): the norm of the result is not wanted -- no finish kernel, and above all NO
// all-reduce (the lazy flush runs at rank-dependent times; a collective there could mismatch across ranks).
// G > 1 (MODE 4 only): the two-coefficient sweep with G column groups per wave and no lane split -- the partner of a sweep 2 run
// as launch_sweep<CPLX, 2, KC / G, NW, G>: same WC and kcw (taken from THAT configuration), so y' is re-formed bit for bit, on
// tiles G times as tall (panel_sweep's G).
template <bool CPLX, int MODE, int KC = (CPLX ? 8 : 16), int NW = (CPLX ? 16 : 8), int SC = 1, int G = 1>
int launch_sweep(lk_context_t c, const double *X, int64_t ldx, int k, double *y, int64_t n, const double *hin,
                 const double *hin2, int store, double *out) {
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    constexpr bool UPDATE = MODE != 1, DOT = MODE <= 2;
    static_assert(KC * NW * SC >= KMAX_FUSED && KC * NW * SC <= KMAX_WIDE, "fused capacity");
    static_assert(G == 1 || (MODE == 4 && SC == 1), "column groups per wave: the two-coefficient sweep only");
    if (k > KC * NW * SC) return fail(LK_ERR_INVALID, "internal: sweep of %d columns on a block that holds %d", k, KC * NW * SC);
    SweepCfg s = sweep_cfg<CPLX, KC / G, NW, SC * G>(c, k, n, MODE == 2 ? c->grid_mult_s2 : (MODE == 4 ? c->grid_mult_s3 : 0));
    if (G > 1) {                             // WC and kcw are the lane-split configuration's; the tiles are G times as tall
        const int64_t tile_rows = (int64_t)(NW / s.WC) * 64 * K<CPLX>::ROWS;
        s.ntiles = (n + tile_rows - 1) / tile_rows;
        int64_t g = (int64_t)c->num_cu * (c->grid_mult_s3 > 0 ? c->grid_mult_s3 : c->grid_mult);
        if (g > s.ntiles) g = s.ntiles;
        if (g > MAX_GRID) g = MAX_GRID;
        if (g < 1) g = 1;
        s.grid = (int)g;
    }
    // ALGORITHMIC bytes of the three-sweep schedule: k+1 | k+2 | k+2 columns
    const double bytes = (double)n * ED * 8.0 * (k + 1 + (UPDATE ? 1 : 0));
    int nblocks = s.grid;
    if (MODE == 1 && c->dot_colwise) {
        // sweep 1 one column at a time (panel_dot_cw): y in registers, U KiB of contiguous rows per wave and column
        auto go = [&](auto uu, int mult) {
            constexpr int UU = decltype(uu)::value;
            const int64_t tile_rows = (int64_t)256 * K<CPLX>::ROWS * UU;
            int64_t g = (n + tile_rows - 1) / tile_rows;
            int64_t cap = (int64_t)c->num_cu * mult;
            if (cap > MAX_GRID) cap = MAX_GRID;
            if (g > cap) {
                // every block the same number of tiles: 977 tiles on 768 blocks would leave a second round for 209 of them
                // (n = 10^6 complex: 5.9 TB/s; on 489 blocks of 2 tiles each 6.9)
                const int64_t rounds = (g + cap - 1) / cap;
                g = (g + rounds - 1) / rounds;
            }
            if (g < 1) g = 1;
            nblocks = (int)g;
            const size_t lds = (size_t)4 * (k + 1) * ED * sizeof(double);
            ProfScope ps(c, "dgs_sweep1", bytes, c->prof_ext);
            if (ps.on && ps.ext)
                hipExtLaunchKernelGGL((panel_dot_cw<CPLX, UU>), dim3(nblocks), dim3(256), lds, c->stream, ps.rec.e0, ps.rec.e1, 0, X, ldx, k,
                                      y, n, c->partial, (int64_t)MAX_GRID, c->guard());
            else
                hipLaunchKernelGGL((panel_dot_cw<CPLX, UU>), dim3(nblocks), dim3(256), lds, c->stream, X, ldx, k, y, n, c->partial,
                                   (int64_t)MAX_GRID, c->guard());
        };
        // long panels: 8 loads per lane and column on 2 blocks per CU (+1-2 % over 4 on 3 at n = 10^8); short ones keep the
        // smaller tile so every CU still gets several tiles
        const bool big = c->cw_u ? c->cw_u == 8 : n >= (int64_t)256 * K<CPLX>::ROWS * 8 * c->num_cu * 2 * 4;
        if (big) go(std::integral_constant<int, 8>{}, c->cw_u ? c->cw_grid_mult : 2);
        else go(std::integral_constant<int, 4>{}, c->cw_grid_mult);
    } else {
        // the sweep's two HIP events ride on the kernel's own dispatch (start / stop timestamps of the launch itself), not on
        // separate stream markers: six markers per Arnoldi step cost 3-4 % of a launch-bound factorisation (n = 10^6 complex)
        ProfScope ps(c, MODE == 1 ? "dgs_sweep1" : (MODE == 2 ? "dgs_sweep2" : "dgs_sweep3"), bytes, c->prof_ext);
        if ((MODE == 3 && c->stream_update) || (MODE == 4 && c->stream_two && G == 1)) {
            ps.ext = false;
            if (ps.on) (void)hipEventRecord(ps.rec.e0, c->stream);
            const int64_t tile_rows = (int64_t)NW * 64 * K<CPLX>::ROWS;
            int64_t g = (n + tile_rows - 1) / tile_rows;
            const int64_t cap = (int64_t)c->num_cu * c->update_grid_mult;
            if (g > cap) g = cap;
            if (g > MAX_GRID) g = MAX_GRID;
            if (g < 1) g = 1;
            nblocks = (int)g;
            hipLaunchKernelGGL((panel_update<CPLX, KC, NW, MODE == 4>), dim3(nblocks), dim3(NW * 64), 0, c->stream, X, ldx, k, y,
                               n, hin, hin2, c->partial, (int64_t)MAX_GRID, c->store_policy, c->guard());
        } else {
            const int st = (store ? (1 | (c->store_policy << 1) | (c->store_split ? 8 : 0)) : 0) | (c->xcd_map ? 16 : 0);
            if (ps.on && ps.ext)
                hipExtLaunchKernelGGL((panel_sweep<CPLX, KC, NW, UPDATE, DOT, MODE == 4, SC, G>), dim3(s.grid), dim3(NW * 64), 0, c->stream,
                                      ps.rec.e0, ps.rec.e1, 0, X, ldx, k, y, n, hin, hin2, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw, st,
                                      c->guard());
            else
                hipLaunchKernelGGL((panel_sweep<CPLX, KC, NW, UPDATE, DOT, MODE == 4, SC, G>), dim3(s.grid), dim3(NW * 64), 0, c->stream, X,
                                   ldx, k, y, n, hin, hin2, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw, st, c->guard());
        }
    }
    HIPCHK(hipGetLastError());
    if (!out) return LK_OK;
    // slots: DOT -> 0..k*ED-1 valid; norm at slot k*ED.  Without DOT only the norm slot is defined.
    const int first = DOT ? 0 : k * ED;
    const int nslots = (k + 1) * ED - first;
    hipLaunchKernelGGL(finish_partials, dim3((nslots + 3) / 4), dim3(256), 0, c->stream,
                       c->partial + (int64_t)first * MAX_GRID, (int64_t)MAX_GRID, nblocks, nslots, out + first);
    HIPCHK(hipGetLastError());
    return allreduce(c, out + first, nslots);
}

template <int MODE>
int sweepm(lk_basis_t Bx, int c0, int k, double *y, const double *hin, const double *hin2, int store, double *out) {
    lk_context_t c = Bx->ctx;
    const double *X = Bx->col(c0);
    if (k > KMAX_FUSED) {
        // wide basis (129..512 columns): the block still holds ALL k columns of its tile, so every sweep is one pass over X.
        // Just beyond 128 columns the REGISTER tile grows instead of the lanes splitting (round 4; "wide_regs", 0 = the round-3
        // lane split everywhere): 8 waves x 32 columns of a full-height 128-row tile for 129..256 real columns, 8 x 24 for
        // 129..192 complex ones -- with the lane split a basis of 129 columns left 7 of a lane group's 16 register columns empty
        // and ran tiles half as tall (5.9 / 4.8 TB/s against 6.7 at k = 128).  Beyond that the 64 lanes of every wave split 2 / 4
        // ways over column groups (8 x 16 columns per group, tiles 1/2 / 1/4 as tall).  The shape depends on k only, so the three
        // sweeps of one DGS share it (sweep 3 re-forms y' in sweep 2's order).
        // (the dot-only sweep, when it runs as a panel_sweep at all -- "dot_colwise" = 0 --, keeps the lane split: its 32-column
        // register tile would spill, and no other sweep has to share its summation order)
        const bool regs = c->wide_regs && MODE != 1;
        // "wide_s3" (round 4): where sweep 2 runs lane-split (SC = 2), the two-coefficient sweep 3 -- no accumulators to keep -- holds
        // BOTH column groups of a wave-column in one wave's registers instead (panel_sweep's G = 2): same WC, kcw and summation
        // order, so y' is re-formed bit for bit, on tiles twice as tall with twice the loads in flight per lane.
        if (Bx->dtype == LK_C128) {
            if (regs && k <= 192) return launch_sweep<true, MODE, 24, 8, 1>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
            if (regs && c->wide_regs >= 2 && k > 2 * KMAX_FUSED && k <= 384) {
                if constexpr (MODE == 4)
                    if (c->wide_s3) return launch_sweep<true, 4, 48, 8, 1, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
                return launch_sweep<true, MODE, 24, 8, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
            }
            if (k <= 2 * KMAX_FUSED) {
                if constexpr (MODE == 4)
                    if (c->wide_s3) return launch_sweep<true, 4, 32, 8, 1, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
                return launch_sweep<true, MODE, 16, 8, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
            }
            return launch_sweep<true, MODE, 16, 8, 4>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
        }
        if (k <= 2 * KMAX_FUSED) {
            if constexpr (MODE != 1)       // (never instantiated for the dot-only sweep: its 32-column register tile spills)
                if (regs) return launch_sweep<false, MODE, 32, 8, 1>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
            if constexpr (MODE == 4)
                if (c->wide_s3) return launch_sweep<false, 4, 32, 8, 1, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
            return launch_sweep<false, MODE, 16, 8, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
        }
        if (regs && c->wide_regs >= 2 && k <= 384) {
            if constexpr (MODE == 4)
                if (c->wide_s3) return launch_sweep<false, 4, 48, 8, 1, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
            return launch_sweep<false, MODE, 24, 8, 2>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
        }
        return launch_sweep<false, MODE, 16, 8, 4>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
    }
    if (Bx->dtype == LK_C128) {
        // complex block shape: 16 waves x 8 columns for narrow bases, 8 waves x 16 columns beyond 32 columns (half the
        // waves per barrier and per LDS exchange: +1-9 % per sweep at k >= 64, A/B in DESIGN.md; "cplx_wide" = threshold, 0 disables).
        // The choice depends on k only, so the three sweeps of one DGS always share it (sweep 3 re-forms y' in sweep 2's order).
        // (the dot-only sweep is free to choose on its own and prefers the narrow shape up to ~56 columns)
        if (c->cplx_wide && k > (MODE == 1 ? c->cplx_wide + 24 : c->cplx_wide)) return launch_sweep<true, MODE, 16, 8>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
        return launch_sweep<true, MODE>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
    }
    // "kc32" (round 4): the real kind's update sweeps on 32-column register tiles (the 129..256-column shape) for narrow bases too:
    // k > kc32 columns run 8 waves x 32 columns (WC = 2 / 4 wave-columns, tiles 2-4 times as tall) -- sweeps 2 and 3 together, they
    // share the summation order.  Measured (profiles/r04_ab_kc32.txt): sweep 2 +2 % and sweep 3 -0.5 % at n = 10^8 (+0.6 % on the three
    // sweeps), null at n = 10^7, -0.6 % at n = 2 10^6 -- so the default (-1) turns it on for k > 32 on LONG panels only, a column
    // beyond the 256 MB memory-side cache (n >= 2^25 rows); 0 = never, v > 0 = for k > v at every size.
    {
        // The size that decides is the GLOBAL row count where the host announced one (lk_set_partition): every rank of a sharded run then
        // picks the same shape whatever its block -- the summation order inside a rank does not depend on the partition, and the 1 / 2 / 4 / 8-GPU
        // points of a scaling run compare the same kernels (round-4 advisor).  Only the DGS pair (MODE 2 + 4) takes it: the single-set
        // update (MODE 3: lazy path, one-pass orthogonalisation) keeps the 16-column tile its A/B was recorded on.
        const int64_t nref = c->n_global > 0 ? c->n_global : Bx->n;
        const int kc32 = c->kc32 >= 0 ? c->kc32 : (nref >= ((int64_t)1 << 25) ? 32 : 0);
        if constexpr (MODE == 2 || MODE == 4)
            if (kc32 && k > kc32) return launch_sweep<false, MODE, 32, 8, 1>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
    }
    return launch_sweep<false, MODE>(c, X, Bx->ld, k, y, Bx->n, hin, hin2, store, out);
}

// M(:, q) = X(:, c0:c0+k)^H Y(:, jy0+q), q < pn <= 4 (k <= KMAX_FUSED), in ONE pass over X; results land in c->red as
// [q][k+1][ED] (slot k of each q = ||Y_q||^2), all-reduced.  pn <= 2: one launch, every wave keeps 16 / 8 columns x 2
// right-hand sides in registers; pn = 3, 4: 4 columns x 4 right-hand sides per wave, column panels of 64.
int dots_p(lk_basis_t Bx, int c0, int k, lk_basis_t By, int jy0, int pn, double *out = nullptr) {
    lk_context_t c = Bx->ctx;
    if (!out) out = c->red;
    const bool cp = Bx->dtype == LK_C128;
    const int ED = Bx->ed();
    const int P = pn <= 2 ? 2 : 4;
    const int nslots = P * (k + 1) * ED;
    int grid = 1;
    {
        ProfScope ps(c, "dots_p", (double)Bx->n * ED * 8.0 * (k + pn));
        if (P == 2) {
            SweepCfg s = cp ? sweep_cfg<true, 8, 16>(c, k, Bx->n) : sweep_cfg<false, 16, 8>(c, k, Bx->n);
            grid = s.grid;
            if (cp)
                hipLaunchKernelGGL((panel_dot_p<true, 8, 16, 2>), dim3(s.grid), dim3(1024), 0, c->stream, Bx->col(c0), Bx->ld, k,
                                   By->col(jy0), By->ld, pn, Bx->n, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw, k + 1, 0, 1);
            else
                hipLaunchKernelGGL((panel_dot_p<false, 16, 8, 2>), dim3(s.grid), dim3(512), 0, c->stream, Bx->col(c0), Bx->ld, k,
                                   By->col(jy0), By->ld, pn, Bx->n, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw, k + 1, 0, 1);
        } else {
            constexpr int PANEL = 64;                                  // 16 waves x 4 columns
            auto cfg = [&](int kk) { return cp ? sweep_cfg<true, 4, 16>(c, kk, Bx->n) : sweep_cfg<false, 4, 16>(c, kk, Bx->n); };
            // every panel launch uses the SAME block count (the smallest any panel wants: the kernels are grid-stride)
            // so that one finish kernel sums every slot over the same blocks
            grid = MAX_GRID;
            for (int j0 = 0; j0 < k; j0 += PANEL) {
                const int g = cfg((k - j0) < PANEL ? (k - j0) : PANEL).grid;
                if (g < grid) grid = g;
            }
            for (int j0 = 0; j0 < k; j0 += PANEL) {
                const int kk = (k - j0) < PANEL ? (k - j0) : PANEL;
                const SweepCfg s = cfg(kk);
                if (cp)
                    hipLaunchKernelGGL((panel_dot_p<true, 4, 16, 4>), dim3(grid), dim3(1024), 0, c->stream, Bx->col(c0 + j0), Bx->ld, kk,
                                       By->col(jy0), By->ld, pn, Bx->n, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw, k + 1, j0, j0 == 0);
                else
                    hipLaunchKernelGGL((panel_dot_p<false, 4, 16, 4>), dim3(grid), dim3(1024), 0, c->stream, Bx->col(c0 + j0), Bx->ld, kk,
                                       By->col(jy0), By->ld, pn, Bx->n, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw, k + 1, j0, j0 == 0);
            }
        }
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(finish_partials, dim3((nslots + 3) / 4), dim3(256), 0, c->stream, c->partial, (int64_t)MAX_GRID, grid,
                       nslots, out);
    HIPCHK(hipGetLastError());
    return allreduce(c, out, nslots);
}

// Passes B and C of the fused block DGS (panel_sweep_p) for columns [jy0, jy0 + pn) of Y against X(:, :k):
//   B: Y' = Y - X H1 in registers, H2 = X^H Y' and ||Y'_q||^2 -> out2 (panel_dot_p's layout, all-reduced)
//   C: Y'' = (Y - X H1) - X H2 stored                      (no reduction: lk_dgs_block returns no norm of Y'')
template <bool CPLX, int KC, int NW, int P>
int block_sweeps(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int pn, const double *out1, double *out2) {
    lk_context_t c = Bx->ctx;
    constexpr int ED = K<CPLX>::ELEM_DOUBLES;
    const SweepCfg s = sweep_cfg<CPLX, KC, NW>(c, k, Bx->n);
    const int nslots = P * (k + 1) * ED;
    {
        ProfScope ps(c, "dgs_block_sweep2", (double)Bx->n * ED * 8.0 * (k + pn));
        hipLaunchKernelGGL((panel_sweep_p<CPLX, KC, NW, P, true, false>), dim3(s.grid), dim3(NW * 64), 0, c->stream, Bx->col(0), Bx->ld, k,
                           By->col(jy0), By->ld, pn, Bx->n, out1, nullptr, k + 1, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw, 0, c->guard());
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(finish_partials, dim3((nslots + 3) / 4), dim3(256), 0, c->stream, c->partial, (int64_t)MAX_GRID, s.grid, nslots, out2);
    HIPCHK(hipGetLastError());
    LKCHK(allreduce(c, out2, nslots));
    {
        ProfScope ps(c, "dgs_block_sweep3", (double)Bx->n * ED * 8.0 * (k + 2 * pn));
        hipLaunchKernelGGL((panel_sweep_p<CPLX, KC, NW, P, false, true>), dim3(s.grid), dim3(NW * 64), 0, c->stream, Bx->col(0), Bx->ld, k,
                           By->col(jy0), By->ld, pn, Bx->n, out1, out2, k + 1, c->partial, (int64_t)MAX_GRID, s.WC, s.kcw,
                           1 | (c->store_policy << 1), c->guard());
    }
    HIPCHK(hipGetLastError());
    return LK_OK;
}

int ensure_scratch(lk_context_t c, int64_t doubles) {
    if (c->scratch_n >= doubles) return LK_OK;
    if (c->scratch) HIPCHK(hipFree(c->scratch));
    c->scratch = nullptr;
    c->scratch_n = 0;
    HIPCHK(hipMalloc((void **)&c->scratch, (size_t)doubles * sizeof(double)));
    c->scratch_n = doubles;
    return LK_OK;
}

// ---- X^H Y on the matrix cores (panel_xhy_mfma) ---------------------------------------------------------------
constexpr int XHY_MIN_P = 5;        // fewer right-hand sides: panel_dot_p (<= 4 per pass) reads X once as well
constexpr int XHY_MAX = 128;        // columns of X and of Y per launch
constexpr int XHY_GROUP = 32;       // right-hand sides per pass of the block DGS (beyond ~32 the pass turns MFMA-bound)

// M = X(:, c0 : c0+k)^H Y(:, jy0 : jy0+p), k, p <= XHY_MAX, into result section `sec` (0 / 1) of c->xhy in panel_dot_p's
// layout [q][k+1][ED] (slot k = ||Y_q||^2), all-reduced.  flags: see the kernel (1 = Y is X, 2 = upper tiles only).
// `slot`: where in the result area the matrix lands, in units of XHY_SLOT doubles (= one group of XHY_GROUP right-hand sides against
// XHY_MAX columns): 0 and XHY_SLOTS / 2 are the two classic sections; the block DGS of a basis wider than XHY_MAX columns keeps one
// slot per column panel of X for H1 (0..3) and one for H2 (4..7).  `may_grow` = false: the workspace already holds coefficients of an
// earlier pass and must not be re-allocated (the caller sized it with the first pass).
constexpr int XHY_SLOTS = 8;
constexpr int64_t XHY_SLOT = (int64_t)XHY_MAX * (XHY_MAX + 1) * 2 * 2 / XHY_SLOTS;   // = XHY_GROUP * (XHY_MAX + 1) * 2 doubles
static_assert(XHY_SLOT == (int64_t)XHY_GROUP * (XHY_MAX + 1) * 2, "eight groups of 32 right-hand sides fill the two result sections");
int dots_mfma(lk_basis_t Bx, int c0, int k, lk_basis_t By, int jy0, int p, int flags, int slot, double **out_dev, bool may_grow = true) {
    lk_context_t c = Bx->ctx;
    const bool cp = Bx->dtype == LK_C128;
    const int ED = Bx->ed();
    const int KP = (k + 15) / 16, PJ = (p + 15) / 16;
    const int NI = KP <= 1 ? 1 : (KP == 2 ? 2 : (KP <= 4 ? 4 : 8));
    const int WR = 8 / NI;
    const int64_t nslots = (int64_t)p * (k + 1) * ED;
    const int64_t sect = (int64_t)XHY_MAX * (XHY_MAX + 1) * 2;                  // doubles per result section
    const int64_t npart_n = (int64_t)c->num_cu * 4 * XHY_MAX;
    // <= 32 right-hand sides: 32-row tiles and a quarter of the accumulators -- 40 KB of LDS and 66-88 VGPRs, so several blocks
    // share a CU and cover each other's barriers and load latency (n = 10^7 real, k = 128, p = 16: 2.54 -> 1.92 ms on 3 blocks
    // per CU, one pass over X at 6.5 TB/s being 1.77; complex 3.73 -> 2.29 ms on 2)
    const bool small = c->xhy_small && PJ <= 2;
    const int TR = (small || cp) ? 32 : 64;     // (complex, more than 32 right-hand sides: 32-row tiles too -- sixteen staged chunks beside 2 x 8 accumulators spilled 68 B)
    const int64_t ntiles = (Bx->n * ED + TR - 1) / TR;
    int64_t g = (int64_t)c->num_cu * (small ? (c->xhy_grid_mult ? c->xhy_grid_mult : (cp ? 2 : 3)) : 1);
    if (g > ntiles) g = ntiles;
    if (g < 1) g = 1;
    const int grid = (int)g, nvb = grid * WR;
    // the workspace: two result sections | norm partials | `blocks` partial result blocks.  It grows only while `may_grow` (a later pass of the block step finds the
    // coefficients of the earlier ones in it), and then at once to what the fused block pass needs (two partial blocks per CU), so that no pass has to grow it again
    auto ensure = [&](int64_t blocks) -> int {
        int64_t need = 2 * sect + npart_n + blocks * nslots;
        if (c->xhy_n >= need) return LK_OK;
        if (!may_grow)
            return fail(LK_ERR_INVALID, "internal: xhy workspace too small for a later pass of the block Gram-Schmidt (%lld < %lld)", (long long)c->xhy_n, (long long)need);
        const int64_t fused = 2 * sect + npart_n + (int64_t)c->num_cu * 2 * nslots;
        if (need < fused) need = fused;
        if (c->xhy) HIPCHK(hipFree(c->xhy));
        c->xhy = nullptr;
        c->xhy_n = 0;
        HIPCHK(hipMalloc((void **)&c->xhy, (size_t)need * sizeof(double)));
        c->xhy_n = need;
        return LK_OK;
    };
    // complex Gram matrix of 5..112 columns: panel_gram_rs's row split and LDS-DMA tiles with three real products per complex one (panel_gram_rs3m / rs3m4, round 6)
    if (cp && c->gemm_3m && flags == 3 && KP <= 7 && c->gram_rs > 0) {
        const int resident = KP == 1 ? 4 : (KP == 2 ? 3 : (KP == 3 ? 2 : 1));                     // blocks per CU (three accumulators per tile: registers)
        const int nbuf = KP == 1 ? 8 : (KP == 2 ? 5 : (KP == 3 ? 3 : 5));                         // (narrow tiles: a deeper ring; 84..140 KB at one block per CU)
        const int64_t nt16 = (Bx->n + 15) / 16;
        int64_t gg = (int64_t)c->num_cu * (c->gram_rs == 1 ? resident : c->gram_rs);
        if (gg > nt16) gg = nt16;
        if (gg < 1) gg = 1;
        LKCHK(ensure(gg));
        double *outg = c->xhy + (int64_t)slot * XHY_SLOT, *npartg = c->xhy + 2 * sect, *partg = npartg + npart_n;
        const size_t ldsg = (size_t)nbuf * KP * 4096;
        {
            ProfScope ps(c, "xhy_mfma", (double)Bx->n * ED * 8.0 * k);
            auto go = [&](auto kern) -> int {
                if (ldsg > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg));
                hipLaunchKernelGGL(kern, dim3((unsigned)gg), dim3(512), ldsg, c->stream, (const double *)Bx->col(c0), Bx->ld, k, Bx->n, partg);
                return LK_OK;
            };
            if (KP == 1) LKCHK(go(&panel_gram_rs3m<1, 8, 8>));
            else if (KP == 2) LKCHK(go(&panel_gram_rs3m<2, 5, 6>));
            else if (KP == 3) LKCHK(go(&panel_gram_rs3m<3, 3, 4>));
            else if (KP == 4) LKCHK(go(&panel_gram_rs3m<4, 5, 2>));
            else if (KP == 5) LKCHK(go(&panel_gram_rs3m<5, 5, 2>));
            else if (KP == 6) LKCHK(go(&panel_gram_rs3m4<6, 5, 2>));                              // four groups of two waves: a quarter of the tile list each
            else LKCHK(go(&panel_gram_rs3m4<7, 5, 2>));
        }
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(finish_xhy, dim3((unsigned)((nslots + 15) / 16)), dim3(256), 0, c->stream, partg, (int)gg, npartg, (int)gg, k, p, ED, flags, outg);
        HIPCHK(hipGetLastError());
        if (out_dev) *out_dev = outg;
        return allreduce(c, outg, nslots);
    }
    // complex Gram matrix beyond 32 columns: upper tiles dealt to the waves, three real products per complex one (panel_gram_mfma3m)
    if (cp && !small && c->gemm_3m && flags == 3) {
        const int64_t nt32 = (Bx->n + 31) / 32;
        int64_t gg = (int64_t)c->num_cu;
        if (gg > nt32) gg = nt32;
        if (gg < 1) gg = 1;
        LKCHK(ensure(gg));
        double *out3 = c->xhy + (int64_t)slot * XHY_SLOT, *npart3 = c->xhy + 2 * sect, *part3 = npart3 + npart_n;
        const size_t lds3 = (size_t)KP * 16 * 34 * 2 * sizeof(double);
        {
            ProfScope ps(c, "xhy_mfma", (double)Bx->n * ED * 8.0 * k);
            if (lds3 > 48 * 1024)
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gram_mfma3m), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            hipLaunchKernelGGL(panel_gram_mfma3m, dim3((unsigned)gg), dim3(512), lds3, c->stream, (const double *)Bx->col(c0), Bx->ld, k, Bx->n, part3);
        }
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(finish_xhy, dim3((unsigned)((nslots + 15) / 16)), dim3(256), 0, c->stream, part3, (int)gg, npart3, (int)gg, k, p, ED, flags, out3);
        HIPCHK(hipGetLastError());
        if (out_dev) *out_dev = out3;
        return allreduce(c, out3, nslots);
    }
    // real Gram matrix of 5..128 columns: the rows of the staged tile dealt to the waves, tiles staged by LDS-DMA (panel_gram_rs, round 6)
    if (!cp && flags == 3 && k <= 128 && c->gram_rs > 0) {
        const int resident = KP <= 3 ? 4 : (KP == 4 ? 3 : (KP <= 6 ? 2 : 1));                    // blocks per CU (LDS ring; beyond 96 columns the accumulators)
        const int nbuf = KP == 1 ? 8 : (KP == 2 ? 5 : (KP <= 6 ? 3 : 4));                        // (narrow tiles: a deeper ring, for the bytes in flight)
        const int64_t nt32 = (Bx->n + 31) / 32;
        int64_t gg = (int64_t)c->num_cu * (c->gram_rs == 1 ? resident : c->gram_rs);
        if (gg > nt32) gg = nt32;
        if (gg < 1) gg = 1;
        LKCHK(ensure(gg));
        double *outg = c->xhy + (int64_t)slot * XHY_SLOT, *npartg = c->xhy + 2 * sect, *partg = npartg + npart_n;
        const size_t ldsg = (size_t)nbuf * KP * 4096;
        {
            ProfScope ps(c, "xhy_mfma", (double)Bx->n * 8.0 * k);
            auto go = [&](auto kern) -> int {
                if (ldsg > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg));
                hipLaunchKernelGGL(kern, dim3((unsigned)gg), dim3(512), ldsg, c->stream, (const double *)Bx->col(c0), Bx->ld, k, Bx->n, partg);
                return LK_OK;
            };
            switch (KP) {                                                                       // <column blocks, ring buffers, waves per SIMD>
            case 1: LKCHK(go(&panel_gram_rs<1, 8, 8>)); break;
            case 2: LKCHK(go(&panel_gram_rs<2, 5, 8>)); break;
            case 3: LKCHK(go(&panel_gram_rs<3, 3, 8>)); break;
            case 4: LKCHK(go(&panel_gram_rs<4, 3, 6>)); break;
            case 5: LKCHK(go(&panel_gram_rs<5, 3, 4>)); break;
            case 6: LKCHK(go(&panel_gram_rs<6, 3, 4>)); break;
            case 7: LKCHK(go(&panel_gram_rs<7, 4, 2>)); break;
            default: LKCHK(go(&panel_gram_rs<8, 4, 2>)); break;
            }
        }
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(finish_xhy, dim3((unsigned)((nslots + 15) / 16)), dim3(256), 0, c->stream, partg, (int)gg, npartg, (int)gg, k, p, ED, flags, outg);
        HIPCHK(hipGetLastError());
        if (out_dev) *out_dev = outg;
        return allreduce(c, outg, nslots);
    }
    LKCHK(ensure(nvb));
    double *out = c->xhy + (int64_t)slot * XHY_SLOT, *npart = c->xhy + 2 * sect, *part = npart + npart_n;
    // complex kind, <= 32 right-hand sides: three real products per complex one on separate real / imaginary planes ("gemm_3m")
    const bool three = cp && small && c->gemm_3m && !(flags & 1);
    size_t lds = three ? (size_t)(KP + PJ) * 16 * 18 * 2 * sizeof(double)
                       : (size_t)(KP + ((flags & 1) ? 0 : PJ)) * 16 * (TR + 2) * sizeof(double);
    // double-buffered tile (one barrier per tile, staging under the MFMAs): the big variants when two buffers fit the 160 KB of a CU
    // ("xhy_db": 1 = the 128-column variants, 2 = the <= 32 right-hand-side variants too, 0 = never)
    const bool db = !three && c->xhy_db && (small ? c->xhy_db >= 2 : true) && 2 * lds <= (size_t)160 * 1024;
    if (db) lds *= 2;
    {
        ProfScope ps(c, "xhy_mfma", (double)Bx->n * ED * 8.0 * (k + ((flags & 1) ? 0 : p)));
        auto go = [&](auto kern) -> int {
            if (lds > 48 * 1024)
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, c->stream, (const double *)Bx->col(c0), Bx->ld, k, (const double *)By->col(jy0), By->ld,
                               p, Bx->n, flags | (c->xhy_debug << 4), NI, part, npart);
            return LK_OK;
        };
        // variant = (kind, <= 32 | <= 128 right-hand sides, rows per tile, double-buffered)
        auto pick = [&](auto cplx) -> int {
            constexpr bool CP = decltype(cplx)::value;
            if (small) return db ? go(&panel_xhy_mfma<CP, 2, 32, true>) : go(&panel_xhy_mfma<CP, 2, 32, false>);
            if constexpr (CP) return db ? go(&panel_xhy_mfma<true, 8, 32, true>) : go(&panel_xhy_mfma<true, 8, 32, false>);
            else return db ? go(&panel_xhy_mfma<false, 8, 64, true>) : go(&panel_xhy_mfma<false, 8, 64, false>);
        };
        if (three) LKCHK(go(&panel_xhy_mfma3m));
        else if (cp) LKCHK(pick(std::true_type{}));
        else LKCHK(pick(std::false_type{}));
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(finish_xhy, dim3((unsigned)((nslots + 15) / 16)), dim3(256), 0, c->stream, part, nvb, npart, grid, k, p, ED, flags, out);
    HIPCHK(hipGetLastError());
    if (out_dev) *out_dev = out;
    return allreduce(c, out, nslots);
}

// Pass B of the block DGS with many right-hand sides, fused (panel_xhy_upd_mfma): Y(:, jy0 : jy0+p) -= X(:, :k) H1, stored, and
// M2 = X^H Y', ||Y'_q||^2 into result section `sec` of c->xhy (panel_dot_p's layout, all-reduced).  k <= 128, p <= 32.
int upd_dots_mfma(lk_basis_t Bx, int c0, int k, lk_basis_t By, int jy0, int p, const double *H1dev, int slot, double **out_dev) {
    lk_context_t c = Bx->ctx;
    const bool cp = Bx->dtype == LK_C128;
    const int ED = Bx->ed();
    const int KG = (k + 31) / 32;
    const int64_t nslots = (int64_t)p * (k + 1) * ED;
    const int64_t sect = (int64_t)XHY_MAX * (XHY_MAX + 1) * 2;
    const int64_t npart_n = (int64_t)c->num_cu * 4 * XHY_MAX;
    const int64_t ntiles = (Bx->n * ED + 31) / 32;
    int64_t g = (int64_t)c->num_cu * (cp ? 1 : 2);           // 78 KB (real) / 113 KB (complex) of LDS at k = 128: two / one blocks of 4 waves per CU
    if (g > ntiles) g = ntiles;
    if (g < 1) g = 1;
    const int grid = (int)g;
    const int64_t need = 2 * sect + npart_n + (int64_t)grid * nslots;
    if (c->xhy_n < need) {
        // the coefficients of pass A live in this buffer: grow it BEFORE pass A ran (lk_dgs_block sizes it up front), never here
        return fail(LK_ERR_INVALID, "internal: xhy workspace too small for the fused block pass (%lld < %lld)", (long long)c->xhy_n, (long long)need);
    }
    double *out = c->xhy + (int64_t)slot * XHY_SLOT, *npart = c->xhy + 2 * sect, *part = npart + npart_n;
    // real kind, 17..32 right-hand sides: row-owner waves on LDS-DMA tiles (panel_xhy_upd_rs, round 6): one block of two four-wave teams per CU, two partial blocks per block
    if (!cp && p > 16 && p <= 32 && c->upd_rs) {
        const int KP = (k + 15) / 16;
        int64_t gr = c->num_cu;
        if (gr > ntiles) gr = ntiles;
        if (gr < 1) gr = 1;
        if (c->xhy_n < 2 * sect + npart_n + 2 * gr * nslots)
            return fail(LK_ERR_INVALID, "internal: xhy workspace too small for the fused block pass (%lld)", (long long)c->xhy_n);
        const size_t ldsr = (size_t)4 * (KP * 4096 + 8192);
        {
            ProfScope ps(c, "xhy_upd_mfma", (double)Bx->n * 8.0 * (k + 2 * p));
            auto go = [&](auto kern) -> int {
                if (ldsr > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr));
                hipLaunchKernelGGL(kern, dim3((unsigned)gr), dim3(512), ldsr, c->stream, (const double *)Bx->col(c0), Bx->ld, k, By->col(jy0), By->ld, p, Bx->n, H1dev, part,
                                   npart, c->guard());
                return LK_OK;
            };
            switch (KP) {
            case 1: LKCHK(go(&panel_xhy_upd_rs<1>)); break;
            case 2: LKCHK(go(&panel_xhy_upd_rs<2>)); break;
            case 3: LKCHK(go(&panel_xhy_upd_rs<3>)); break;
            case 4: LKCHK(go(&panel_xhy_upd_rs<4>)); break;
            case 5: LKCHK(go(&panel_xhy_upd_rs<5>)); break;
            case 6: LKCHK(go(&panel_xhy_upd_rs<6>)); break;
            case 7: LKCHK(go(&panel_xhy_upd_rs<7>)); break;
            default: LKCHK(go(&panel_xhy_upd_rs<8>)); break;
            }
        }
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(finish_xhy, dim3((unsigned)((nslots + 15) / 16)), dim3(256), 0, c->stream, part, (int)(2 * gr), npart, (int)(2 * gr), k, p, ED, 0, out);
        HIPCHK(hipGetLastError());
        if (out_dev) *out_dev = out;
        return allreduce(c, out, nslots);
    }
    const size_t lds = (size_t)(KG * 32 * 34 + 32 * 34 + KG * 32 * 34 * (cp ? 2 : 1)) * sizeof(double);
    {
        ProfScope ps(c, "xhy_upd_mfma", (double)Bx->n * ED * 8.0 * (k + 2 * p));
        auto go = [&](auto kern) -> int {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, c->stream, (const double *)Bx->col(c0), Bx->ld, k, By->col(jy0), By->ld, p, Bx->n,
                               H1dev, part, npart, c->gemm_store_policy | (c->upd_debug << 4), c->guard());
            return LK_OK;
        };
        if (cp) LKCHK(go(&panel_xhy_upd_mfma<true>));
        else LKCHK(go(&panel_xhy_upd_mfma<false>));
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(finish_xhy, dim3((unsigned)((nslots + 15) / 16)), dim3(256), 0, c->stream, part, grid, npart, grid, k, p, ED, 0, out);
    HIPCHK(hipGetLastError());
    if (out_dev) *out_dev = out;
    return allreduce(c, out, nslots);
}

// ---- tall-skinny product launcher (panel_gemm) -------------------------------------------------------------
constexpr int GEMM_QB = 16;       // accumulators (output columns) per lane
constexpr int GEMM_NQG = 4;       // output-column groups per block (= waves per block)

inline int64_t gemm_packed_doubles(int k, int q, int ED) {
    const int64_t valu = (int64_t)((q + GEMM_QB - 1) / GEMM_QB) * k * GEMM_QB * ED;
    const int QBm = ED == 2 ? 8 : 16;                                  // MFMA layout: [groups][k/4][NA][64]
    const int64_t mfma = (int64_t)((q + QBm - 1) / QBm + 8) * ((KMAX_FUSED + 3) / 4) * 64;   // +8: a launch stages NG whole groups
    return valu > mfma ? valu : mfma;
}

// Y(:, jy0 : jy0+q) (+)= sign * X(:, c0 : c0+k) * C, C = DEVICE coefficients, column-major k x q with leading dimension
// ldc (elements).  `pack` = device workspace of gemm_packed_doubles(k, q, ED) doubles.  One pass over X per 64 outputs.
int gemm_launch_valu(lk_basis_t Bx, int c0, int k, lk_basis_t By, int jy0, int q, const double *Cdev, int64_t ldc, double sign,
                     int accumulate, double *pack) {
    lk_context_t c = Bx->ctx;
    const bool cp = Bx->dtype == LK_C128;
    const int ED = Bx->ed();
    // accumulators per lane: the smallest of 1 / 2 / 4 / 8 / 16 that holds the product (q <= 16), 16 beyond
    const int QB = q <= 1 ? 1 : (q <= 2 ? 2 : (q <= 4 ? 4 : (q <= 8 ? 8 : GEMM_QB)));
    const int total = ((q + QB - 1) / QB) * k * QB * ED;
    hipLaunchKernelGGL(pack_coef, dim3((total + 255) / 256 > 64 ? 64 : (total + 255) / 256), dim3(256), 0, c->stream, Cdev, ldc, k, q,
                       QB, ED, sign, pack);
    HIPCHK(hipGetLastError());
    for (int q0 = 0; q0 < q; q0 += QB * GEMM_NQG) {
        const int qn = (q - q0) < QB * GEMM_NQG ? (q - q0) : QB * GEMM_NQG;
        const int groups = (qn + QB - 1) / QB;
        const int QGB = groups <= 1 ? 1 : (groups == 2 ? 2 : 4);
        const int64_t tile_rows = (int64_t)(4 / QGB) * 64 * (cp ? 1 : 2);
        int64_t g = (Bx->n + tile_rows - 1) / tile_rows;
        const int64_t cap = (int64_t)c->num_cu * c->gemm_grid_mult;
        if (g > cap) g = cap;
        if (g < 1) g = 1;
        const double *Cp = pack + (int64_t)(q0 / QB) * k * QB * ED;
        ProfScope ps(c, "lincomb", (double)Bx->n * ED * 8.0 * (k + qn * (accumulate ? 2 : 1)));
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(256), 0, c->stream, (const double *)Bx->col(c0), Bx->ld, k, By->col(jy0 + q0), By->ld, qn,
                               Cp, Bx->n, accumulate, QGB, c->gemm_store_policy, c->guard());
        };
        // columns in flight per lane: 16 (real) / 8 (complex) loads of 16 B for the narrow shapes, half that beside 8-16 accumulators
        if (cp) {
            switch (QB) {
            case 1: go(&panel_gemm<true, 8, 1>); break;
            case 2: go(&panel_gemm<true, 8, 2>); break;
            case 4: go(&panel_gemm<true, 8, 4>); break;
            case 8: go(&panel_gemm<true, 4, 8>); break;
            default: go(&panel_gemm<true, 4, GEMM_QB>);
            }
        } else {
            switch (QB) {
            case 1: go(&panel_gemm<false, 16, 1>); break;
            case 2: go(&panel_gemm<false, 16, 2>); break;
            case 4: go(&panel_gemm<false, 16, 4>); break;
            case 8: go(&panel_gemm<false, 8, 8>); break;
            default: go(&panel_gemm<false, 8, GEMM_QB>);
            }
        }
        HIPCHK(hipGetLastError());
    }
    return LK_OK;
}

// MFMA path: k-chunks of 128 columns (one LDS image of the coefficient tiles per launch), NG output groups per launch
// (64 output columns per pass over X for either kind).
template <bool CPLX, int NG>
int gemm_mfma_one(lk_context_t c, const double *X, int64_t ldx, int kk, double *Y, int64_t ldy, int qn, const double *Cp, int64_t n,
                  int accumulate) {
    const int nt = (kk + 3) / 4;
    const size_t lds = (size_t)NG * nt * 64 * sizeof(double);
    if (lds > 48 * 1024)                         // more than the default dynamic-LDS limit needs the opt-in (gfx950: 160 KB per workgroup)
    {
        if constexpr (!(CPLX && NG >= 8))
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gemm_mfma<CPLX, NG, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gemm_mfma<CPLX, NG, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    constexpr int tile_rows = 8 * 2 * (CPLX ? 16 : 32);
    int64_t g = (n + tile_rows - 1) / tile_rows;
    const int64_t cap = (int64_t)c->num_cu * (NG >= 8 ? 1 : (NG >= 4 && CPLX ? 2 : c->gemm_grid_mult));
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    if constexpr (!CPLX && NG <= 2) {
        if (accumulate && c->gemm_prefetch_y) {          // the accumulating update of the block Gram-Schmidt: Y's tile loaded ahead of the k-loop
            if (lds > 48 * 1024)
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gemm_mfma<CPLX, NG, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            // (the rolling prefetch of X beside the prefetched tile of Y does not fit the register file: batch schedule here)
            hipLaunchKernelGGL((panel_gemm_mfma<CPLX, NG, true, false>), dim3((unsigned)g), dim3(512), lds, c->stream, X, ldx, kk, Y, ldy, qn, Cp, n, accumulate,
                               c->gemm_store_policy, c->guard());
            HIPCHK(hipGetLastError());
            return LK_OK;
        }
    }
    constexpr bool CAN_ROLL = !(CPLX && NG >= 8);          // (64 complex outputs as the doubled real problem: the ring does not fit the register file)
    bool rolled = false;
    if constexpr (CAN_ROLL) {
        // "gemm_roll" = 1 (default): the real product with 33..64 outputs per pass -- the restart update X <- X Z of krylov_schur -- on the straight-line
        // ring (k = 128, q = 64 at n = 10^7: 3.45-3.73 -> 3.16-3.24 ms, 51 TFLOP/s); the narrower ones measured the same on either schedule
        // (k = 64, q = 32: 1.38-1.44 ms both) and keep the batch schedule's three blocks per CU; 2 = every variant that has a ring
        // (the real kind's straight-line ring is unrolled for the basis widths 128 and 64: any other width keeps the batch schedule)
        if ((CPLX || kk == 128 || kk == 64) && (c->gemm_roll >= 2 || (c->gemm_roll == 1 && !CPLX && NG == 4))) {
            hipLaunchKernelGGL((panel_gemm_mfma<CPLX, NG, false, true>), dim3((unsigned)g), dim3(512), lds, c->stream, X, ldx, kk, Y, ldy, qn, Cp, n, accumulate,
                               c->gemm_store_policy, c->guard());
            rolled = true;
        }
    }
    if (!rolled)
        hipLaunchKernelGGL((panel_gemm_mfma<CPLX, NG, false, false>), dim3((unsigned)g), dim3(512), lds, c->stream, X, ldx, kk, Y, ldy, qn, Cp, n, accumulate,
                           c->gemm_store_policy, c->guard());
    HIPCHK(hipGetLastError());
    return LK_OK;
}

// complex kind, three real products per complex one (panel_gemm_mfma3m): groups of 16 complex outputs, up to 4 per launch
template <int NG, int NR = (NG >= 4 ? 1 : 2)>
int gemm_mfma3m_one(lk_context_t c, const double *X, int64_t ldx, int kk, double *Y, int64_t ldy, int qn, const double *Cp, int64_t n,
                    int accumulate) {
    const int nt = (kk + 3) / 4;
    const size_t lds = (size_t)NG * nt * 128 * sizeof(double);
    const bool roll = c->gemm_roll >= 1 && (kk == 128 || kk == 64);       // "gemm_roll": see gemm_mfma_one (the complex kind: every width of product measured faster on the ring)
    if (lds > 48 * 1024) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gemm_mfma3m<NG, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gemm_mfma3m<NG, NR, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    constexpr int tile_rows = 8 * NR * 16;
    int64_t g = (n + tile_rows - 1) / tile_rows;
    const int64_t cap = (int64_t)c->num_cu * (NG >= 4 ? 2 : (NG >= 2 ? 2 : c->gemm_grid_mult));
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    if (roll)
        hipLaunchKernelGGL((panel_gemm_mfma3m<NG, NR, true>), dim3((unsigned)g), dim3(512), lds, c->stream, X, ldx, kk, Y, ldy, qn, Cp, n, accumulate,
                           c->gemm_store_policy, c->guard());
    else
        hipLaunchKernelGGL((panel_gemm_mfma3m<NG, NR>), dim3((unsigned)g), dim3(512), lds, c->stream, X, ldx, kk, Y, ldy, qn, Cp, n, accumulate,
                           c->gemm_store_policy, c->guard());
    HIPCHK(hipGetLastError());
    return LK_OK;
}

int gemm_launch_mfma3m(lk_basis_t Bx, int c0, int k, lk_basis_t By, int jy0, int q, const double *Cdev, int64_t ldc, double sign,
                       int accumulate, double *pack) {
    lk_context_t c = Bx->ctx;
    constexpr int QB = 16, NGMAX = 4;
    for (int kc = 0; kc < k; kc += KMAX_FUSED) {
        const int kk = (k - kc) < KMAX_FUSED ? (k - kc) : KMAX_FUSED;
        const int nt = (kk + 3) / 4;
        const int ngroups = (q + QB - 1) / QB;
        const int total = ngroups * nt * 128;
        hipLaunchKernelGGL(pack_coef_mfma3m, dim3((total + 255) / 256 > 64 ? 64 : (total + 255) / 256), dim3(256), 0, c->stream,
                           Cdev + (int64_t)kc * 2, ldc, kk, q, sign, pack);
        HIPCHK(hipGetLastError());
        const int acc = (accumulate || kc > 0) ? 1 : 0;
        for (int g0 = 0; g0 < ngroups; g0 += NGMAX) {
            const int groups = (ngroups - g0) < NGMAX ? (ngroups - g0) : NGMAX;
            const int qn = (q - g0 * QB) < groups * QB ? (q - g0 * QB) : groups * QB;
            const double *Cp = pack + (int64_t)g0 * nt * 128;
            const double *Xp = Bx->col(c0 + kc);
            double *Yp = By->col(jy0 + g0 * QB);
            ProfScope ps(c, "lincomb", (double)Bx->n * 16.0 * (kk + qn * (acc ? 2 : 1)));
            int rc;
            if (groups <= 1) rc = gemm_mfma3m_one<1>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
            else if (groups == 2) rc = gemm_mfma3m_one<2>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
            else rc = gemm_mfma3m_one<4>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
            LKCHK(rc);
        }
    }
    return LK_OK;
}

int gemm_launch_mfma(lk_basis_t Bx, int c0, int k, lk_basis_t By, int jy0, int q, const double *Cdev, int64_t ldc, double sign,
                     int accumulate, double *pack) {
    lk_context_t c = Bx->ctx;
    const bool cp = Bx->dtype == LK_C128;
    const int ED = Bx->ed();
    if (cp && c->gemm_3m) return gemm_launch_mfma3m(Bx, c0, k, By, jy0, q, Cdev, ldc, sign, accumulate, pack);
    const int QB = cp ? 8 : 16, NGMAX = cp ? 8 : 4;
    for (int kc = 0; kc < k; kc += KMAX_FUSED) {
        const int kk = (k - kc) < KMAX_FUSED ? (k - kc) : KMAX_FUSED;
        const int nt = (kk + 3) / 4;
        const int ngroups = (q + QB - 1) / QB;
        const int total = ngroups * nt * 64;
        hipLaunchKernelGGL(pack_coef_mfma, dim3((total + 255) / 256 > 64 ? 64 : (total + 255) / 256), dim3(256), 0, c->stream,
                           Cdev + (int64_t)kc * ED, ldc, kk, q, cp ? 1 : 0, sign, pack);
        HIPCHK(hipGetLastError());
        const int acc = (accumulate || kc > 0) ? 1 : 0;
        for (int g0 = 0; g0 < ngroups; g0 += NGMAX) {
            const int groups = (ngroups - g0) < NGMAX ? (ngroups - g0) : NGMAX;
            const int qn = (q - g0 * QB) < groups * QB ? (q - g0 * QB) : groups * QB;
            const double *Cp = pack + (int64_t)g0 * nt * 64;
            const double *Xp = Bx->col(c0 + kc);
            double *Yp = By->col(jy0 + g0 * QB);
            ProfScope ps(c, "lincomb", (double)Bx->n * ED * 8.0 * (kk + qn * (acc ? 2 : 1)));
            int rc;
            if (cp) {
                if (groups <= 1) rc = gemm_mfma_one<true, 1>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
                else if (groups == 2) rc = gemm_mfma_one<true, 2>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
                else if (groups <= 4) rc = gemm_mfma_one<true, 4>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
                else rc = gemm_mfma_one<true, 8>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
            } else {
                if (groups <= 1) rc = gemm_mfma_one<false, 1>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
                else if (groups == 2) rc = gemm_mfma_one<false, 2>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
                else rc = gemm_mfma_one<false, 4>(c, Xp, Bx->ld, kk, Yp, By->ld, qn, Cp, Bx->n, acc);
            }
            LKCHK(rc);
        }
    }
    return LK_OK;
}

int gemm_launch(lk_basis_t Bx, int c0, int k, lk_basis_t By, int jy0, int q, const double *Cdev, int64_t ldc, double sign,
                int accumulate, double *pack) {
    // narrow products (q < gemm_mfma_min) stream X through the VALU kernel with q accumulators per lane; wider ones go to the
    // matrix cores (crossover from profiles/r03_lincomb.jsonl)
    const int mfma_min = Bx->ctx->gemm_mfma_min > 0 ? Bx->ctx->gemm_mfma_min : (Bx->dtype == LK_C128 ? 9 : 5);
    if (Bx->ctx->gemm_mfma && q >= mfma_min) return gemm_launch_mfma(Bx, c0, k, By, jy0, q, Cdev, ldc, sign, accumulate, pack);
    return gemm_launch_valu(Bx, c0, k, By, jy0, q, Cdev, ldc, sign, accumulate, pack);
}

int dot_device(lk_basis_t Bx, int jx, lk_basis_t By, int jy, double *out_dev) {
    lk_context_t c = Bx->ctx;
    const int64_t nv = Bx->n * Bx->ed() / 2 + 1;
    int g = blas1_grid(c, nv);
    if (g > MAX_GRID) g = MAX_GRID;
    {
        ProfScope ps(c, "blas1", (double)Bx->n * Bx->ed() * 16.0);
        if (Bx->dtype == LK_C128)
            hipLaunchKernelGGL(k_dot<true>, dim3(g), dim3(256), 0, c->stream, Bx->col(jx), By->col(jy), Bx->n, c->partial,
                               (int64_t)MAX_GRID, blas1_nt(Bx));
        else
            hipLaunchKernelGGL(k_dot<false>, dim3(g), dim3(256), 0, c->stream, Bx->col(jx), By->col(jy), Bx->n, c->partial,
                               (int64_t)MAX_GRID, blas1_nt(Bx));
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(finish_partials, dim3(1), dim3(256), 0, c->stream, c->partial, (int64_t)MAX_GRID, g, 2, out_dev);
    HIPCHK(hipGetLastError());
    return allreduce(c, out_dev, 2);
}

int fetch(lk_context_t c, int section0, int nsections, int stride = RED_SECTION) {
    // copy result sections to the pinned mirror and wait
    HIPCHK(hipMemcpyAsync(c->red_host + (size_t)section0 * stride, c->red + (size_t)section0 * stride,
                          (size_t)nsections * stride * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    return LK_OK;
}

int scal_launch(lk_basis_t B, int j, double ar, double ai, const double *inv_sqrt_of, double tol, int *stop_out = nullptr,
                double tol_break = 0.0) {
    lk_context_t c = B->ctx;
    const int64_t nv = B->n * B->ed() / 2 + 1;
    ProfScope ps(c, "blas1", (double)B->n * B->ed() * 16.0);
    if (B->dtype == LK_C128)
        hipLaunchKernelGGL(k_scal<true>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, B->col(j), B->n, ar, ai,
                           inv_sqrt_of, tol, c->guard(), stop_out, tol_break, blas1_nt(B));
    else
        hipLaunchKernelGGL(k_scal<false>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, B->col(j), B->n, ar, ai,
                           inv_sqrt_of, tol, c->guard(), stop_out, tol_break, blas1_nt(B));
    HIPCHK(hipGetLastError());
    return LK_OK;
}

// ---- lazy batching of the per-object path --------------------------------------------------------------
// What an unchanged LightKrylov issues per Gram-Schmidt pass (gram_schmidt.fypp:113-154, AbstractVectors.fypp:571-603):
//   y%norm(); X(1..k)%dot(y); allocate(proj, source=X(1)); proj%zero(); proj%axpby(h_i, X(i), 1) x k; y%sub(proj)
// Deferred here as: virtual zero -> queue (proj = X h, never written) -> sub (y -= X h pending) -> the next pass's
// y%norm() runs ONE sweep that forms y' = y - X h, stores it, and returns ||y'||^2 and X^H y' (memoised for the k dot
// calls that follow): one pass over X per Gram-Schmidt pass, as in the fused lk_dgs.  Every other entry point first
// brings the vectors it touches up to date, so no call ever observes a stale vector.

// h = -(s * a_i) into the device coefficient buffer (the sweeps compute y - X h)
int stage_coef(lk_context_t c, const std::vector<double> &a, int cnt, int ED, double sr, double si) {
    if (c->coef_ev) HIPCHK(hipEventSynchronize(c->coef_ev));      // previous staging copy has left the pinned buffer
    for (int i = 0; i < cnt; ++i) {
        if (ED == 2) {
            const double ar = a[2 * i], ai = a[2 * i + 1];
            c->coef_host[2 * i] = -(sr * ar - si * ai);
            c->coef_host[2 * i + 1] = -(sr * ai + si * ar);
        } else {
            c->coef_host[i] = -(sr * a[i]);
        }
    }
    HIPCHK(hipMemcpyAsync(c->coef, c->coef_host, (size_t)cnt * ED * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->coef_ev, c->stream));
    return LK_OK;
}

// pending y += s X a as one plain panel update (no norm wanted: no finish kernel, no collective)
int apply_sub(lk_context_t c) {
    if (!c->sub.active) return LK_OK;
    auto &q = c->queue;
    c->sub.active = false;
    c->forget_memos();
    LKCHK(stage_coef(c, q.coef, q.cnt, q.Bx->ed(), c->sub.s[0], c->sub.s[1]));
    c->fusion_stats[1] += 1;
    return sweepm<3>(q.Bx, q.j0, q.cnt, c->sub.By->col(c->sub.jy), c->coef, nullptr, 1, nullptr);
}

void drop_queue(lk_context_t c) {
    if (c->queue.active && c->queue.zeroed && c->queue.cnt > 0) c->fusion_stats[2] += 1;
    c->queue.active = false;
}

// write the queue's target: T = [zeroed ? 0 : T] + X a
int materialise_queue(lk_context_t c) {
    auto &q = c->queue;
    if (!q.active) return LK_OK;
    q.active = false;
    if (!q.By) {                                                    // storage already gone: nothing to write
        if (q.zeroed && q.cnt > 0) c->fusion_stats[2] += 1;
        return LK_OK;
    }
    c->forget_memos();                  // a column changes: batched dots may have covered it
    double *T = q.By->col(q.jy);
    if (q.cnt == 1) {
        // a single queued term is a plain axpby (cg's x%axpby(alpha, p, 1), r%axpby(-alpha, Ap, 1): CG.fypp:125-131): the
        // BLAS-1 kernel streams it faster than a one-column panel update; T = a x + (zeroed ? 0 : 1) T
        const bool cp = q.Bx->dtype == LK_C128;
        const int64_t nv = q.Bx->n * q.Bx->ed() / 2 + 1;
        const double b = q.zeroed ? 0.0 : 1.0;
        if (q.zeroed) c->fusion_stats[3] += 1;
        c->lazy_stats[3] += 1;
        ProfScope ps(c, "blas1", (double)q.Bx->n * q.Bx->ed() * 24.0);
        if (cp)
            hipLaunchKernelGGL(k_axpby<true>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, q.coef[0], q.coef[1], q.Bx->col(q.j0), b, 0.0,
                               T, q.Bx->n, blas1_nt(q.Bx));
        else
            hipLaunchKernelGGL(k_axpby<false>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, q.coef[0], 0.0, q.Bx->col(q.j0), b, 0.0, T,
                               q.Bx->n, blas1_nt(q.Bx));
        HIPCHK(hipGetLastError());
        return LK_OK;
    }
    if (q.zeroed) {
        HIPCHK(hipMemsetAsync(T, 0, (size_t)q.By->n * q.By->ed() * sizeof(double), c->stream));
        if (q.cnt > 0) c->fusion_stats[3] += 1;
    }
    if (q.cnt == 0) return LK_OK;
    LKCHK(stage_coef(c, q.coef, q.cnt, q.Bx->ed(), 1.0, 0.0));
    c->lazy_stats[3] += 1;
    return sweepm<3>(q.Bx, q.j0, q.cnt, T, c->coef, nullptr, 1, nullptr);   // norm unused: no collective
}

// Everything pending becomes real (panel-level entry points, sync, tuning changes).
int lazy_flush(lk_context_t c) {
    LKCHK(apply_sub(c));
    return materialise_queue(c);
}

// Called at the top of every ABI entry that works on whole panels (or whose operands are not tracked).
inline int lazy_enter(lk_context_t c, bool mutates) {
    if (!c->lazy) return LK_OK;
    if (mutates) { c->forget_memos(); }
    return lazy_flush(c);
}

// Per-vector entries name their operands: w = the column written (overwrite: its old contents are not read),
// r1 / r2 = columns read.  A virtual T survives the call unless the call reads it, partially updates it, or writes
// into one of the columns it is defined from.
struct VecRef { lk_basis_t B; int j; };
int lazy_enter_vec(lk_context_t c, const VecRef *w, bool overwrite, const VecRef *r1, const VecRef *r2) {
    if (!c->lazy) return LK_OK;
    if (w) { c->forget_memos(); }
    if (c->sub.active && w && overwrite && w->B->col(w->j) == c->sub.By->col(c->sub.jy) &&
        !(r1 && r1->B->col(r1->j) == w->B->col(w->j)) && !(r2 && r2->B->col(r2->j) == w->B->col(w->j)))
        c->sub.active = false;                                      // the vector with the pending update is overwritten: nothing to apply
    LKCHK(apply_sub(c));                                            // y is live: bring it up to date first
    auto &q = c->queue;
    if (!q.active) return LK_OK;
    if (!q.zeroed) return materialise_queue(c);                     // the target's own contents are pending
    const double *T = q.By ? q.By->col(q.jy) : nullptr;
    auto is_T = [&](const VecRef *v) { return v && T && v->B->col(v->j) == T; };
    if (is_T(r1) || is_T(r2)) return materialise_queue(c);
    if (is_T(w)) {
        if (overwrite) { drop_queue(c); return LK_OK; }
        return materialise_queue(c);
    }
    if (w && q.cnt > 0) {                                           // a write into the columns T is defined from
        const double *p = w->B->col(w->j);
        const double *lo = q.Bx->col(q.j0), *hi = lo + (int64_t)q.cnt * q.Bx->ld * q.Bx->ed();
        if (p >= lo && p < hi) return materialise_queue(c);
    }
    return LK_OK;
}

// sub pending on y: ONE sweep forms y' = y + s X a, stores it, and leaves X^H y' and ||y'||^2 in the memos.
int fused_sub_with_dots(lk_context_t c) {
    auto &q = c->queue;
    const int ED = q.Bx->ed();
    double *y = c->sub.By->col(c->sub.jy);
    c->sub.active = false;
    LKCHK(stage_coef(c, q.coef, q.cnt, ED, c->sub.s[0], c->sub.s[1]));
    LKCHK((sweepm<2>(q.Bx, q.j0, q.cnt, y, c->coef, nullptr, 1, c->red)));
    LKCHK(fetch(c, 0, 1, red_stride(q.cnt)));
    auto &mm = c->memo;
    mm.vals.assign(c->red_host, c->red_host + (size_t)q.cnt * ED);
    mm.valid = true; mm.xbase = q.Bx->data; mm.y = y; mm.j0 = q.j0; mm.cnt = q.cnt;
    c->nmemo.valid = true; c->nmemo.y = y; c->nmemo.nrm2 = c->red_host[(size_t)q.cnt * ED];
    c->fusion_stats[0] += 1;
    return LK_OK;
}

// Core of double_gram_schmidt_step for one vector; results stay in c->red (device):
//   section 0: h1[0..k), nrm2(y)    section 1: h2[0..k), nrm2(y')   section 2 (slot k): nrm2(y'')
int dgs_device(lk_basis_t Bx, int k, double *y, bool two_pass, double *red_base = nullptr, int stride = 0, int c0 = 0) {
    lk_context_t c = Bx->ctx;
    double *base = red_base ? red_base : c->red;
    if (!stride) stride = red_stride(k);
    double *r0 = base, *r1 = base + stride, *r2 = base + 2 * stride;
    if (k <= KMAX_WIDE) {
        LKCHK((sweepm<1>(Bx, c0, k, y, nullptr, nullptr, 1, r0)));    // h1 = X^H y ; ||y||^2
        if (two_pass && c->recompute_update) {
            LKCHK((sweepm<2>(Bx, c0, k, y, r0, nullptr, 0, r1)));     // y' = y - X h1 (registers only); h2 = X^H y'; ||y'||^2
            LKCHK((sweepm<4>(Bx, c0, k, y, r0, r1, 1, r2)));          // y'' = (y - X h1) - X h2 ; ||y''||^2
        } else if (two_pass) {
            LKCHK((sweepm<2>(Bx, c0, k, y, r0, nullptr, 1, r1)));     // y' = y - X h1 ; h2 = X^H y' ; ||y'||^2
            LKCHK((sweepm<3>(Bx, c0, k, y, r1, nullptr, 1, r2)));     // y'' = y' - X h2 ; ||y''||^2
        } else {
            LKCHK((sweepm<3>(Bx, c0, k, y, r0, nullptr, 1, r1)));     // y' = y - X h1 ; ||y'||^2
        }
        return LK_OK;
    }
    return fail(LK_ERR_INVALID, "internal: dgs_device called with k=%d > %d", k, KMAX_WIDE);
}

// ---- single-launch step (lk_resident.hip.h) -------------------------------------------------------------------
constexpr int RES_S = RED_SECTION;                   // slot stride of the hand-off buffers: (128 + 1) * 2 doubles
constexpr int RES_MAX_GRID = RES_GRID_CAP;

int resident_ws(lk_context_t c, ResidentWs *ws) {
    if (!c->res_cnt) {
        HIPCHK(hipMalloc((void **)&c->res_cnt, (size_t)RES_CNT_STRIDE * sizeof(unsigned)));
        HIPCHK(hipMemsetAsync(c->res_cnt, 0, (size_t)RES_CNT_STRIDE * sizeof(unsigned), c->stream));
        const size_t gran_bytes = (size_t)RES_EPISODES * (RES_GRID_CAP + RES_GROUPS) * RES_S * 16;
        HIPCHK(hipMalloc(&c->res_gran, gran_bytes));
        HIPCHK(hipMemsetAsync(c->res_gran, 0, gran_bytes, c->stream));          // (tag 0 is never a launch's)
        HIPCHK(hipMalloc((void **)&c->res_tim, 8 * sizeof(long long)));
        HIPCHK(hipMemsetAsync(c->res_tim, 0, 8 * sizeof(long long), c->stream));
    }
    ws->tim = c->res_tim;
    ws->gran = (v2d *)c->res_gran;
    ws->epoch = ++c->res_epoch;
    ws->cnt = c->res_cnt;
    ws->S = RES_S;
    return LK_OK;
}

// a launch that gave up leaves the abort word raised: clear it (stream ordered) and stop trying on this context
int resident_recover(lk_context_t c) {
    c->resident_off = true;
    c->resident_stats[1] += 1;
    c->resident_retry_at = c->resident_fallback_steps + c->resident_pause;
    if (c->resident_pause < ((int64_t)1 << 20)) c->resident_pause *= 2;
    if (c->res_cnt) HIPCHK(hipMemsetAsync(c->res_cnt, 0, (size_t)RES_CNT_STRIDE * sizeof(unsigned), c->stream));
    return LK_OK;
}

// does the two-pass step of k columns against y run as ONE launch?  Only on a single rank (the phases' sums meet inside the
// launch; a sharded run needs the all-reduce between them), for k <= 128, and while the panel fits the memory-side cache.
bool resident_would_apply(lk_basis_t Bx, int k) {
    lk_context_t c = Bx->ctx;
    if (!c->resident || c->nranks > 1 || k < 1 || k > KMAX_FUSED) return false;   // (an all-reduce over ONE rank is the identity)
    const double mb = (double)Bx->n * Bx->ed() * 8.0 * (k + 1) / (1024.0 * 1024.0);
    return mb <= (double)c->resident_max_mb;
}
bool resident_applies(lk_basis_t Bx, int k) { return !Bx->ctx->resident_off && resident_would_apply(Bx, k); }
// a step that the pause sent to the three sweeps counts towards its end
void resident_note_fallback(lk_basis_t Bx, int k) {
    lk_context_t c = Bx->ctx;
    if (c->resident_off && resident_would_apply(Bx, k)) c->resident_fallback_steps += 1;
}
// Called where NO step is in flight between its enqueue and the host's look at its status (the entry of lk_dgs, the start of an asynchronous
// batch): the answer of resident_applies must not change in between.
void resident_maybe_rearm(lk_context_t c) {
    if (c->resident_off && c->resident_fallback_steps >= c->resident_retry_at) c->resident_off = false;
}

int dgs_resident_launch(lk_basis_t Bx, int k, double *y, double *out, int rs, bool normalize, double tol_scale, double tol_break, int *stop_out,
                        int c0 = 0);

// One Gram-Schmidt step + normalise of an ASYNCHRONOUS batch (lk_arnoldi / lk_lanczos / lk_bidiag): results into the step slot, the
// normalise skipped below tol_scale, the device stop flag raised below tol_break.  A cache-resident panel takes the single launch
// (lk_resident.hip.h); a launch that gives up raises the stop flag and leaves status 1 in the slot -- resident_status() below.
int dgs_step_async(lk_basis_t Bx, int k, lk_basis_t By, int jy, double *slot, int rs, double tol_scale, double tol_break, int c0 = 0) {
    lk_context_t c = Bx->ctx;
    const int ED = Bx->ed();
    if (resident_applies(Bx, k)) return dgs_resident_launch(Bx, k, By->col(jy), slot, rs, true, tol_scale, tol_break, c->stop_dev, c0);
    resident_note_fallback(Bx, k);
    LKCHK(dgs_device(Bx, k, By->col(jy), true, slot, rs, c0));
    return scal_launch(By, jy, 1.0, 0.0, slot + 2 * rs + (size_t)k * ED, tol_scale, c->stop_dev, tol_break);
}
// host side of a finished batch: did the step that used `slot` give up (1: redo it on the three-sweep schedule; the context has been
// switched over) or fail (error)?  0 = it ran, or it never was a single launch.
int resident_status(lk_basis_t Bx, int k, const double *slot_host, int rs, int *redo) {
    *redo = 0;
    if (!resident_applies(Bx, k)) return LK_OK;
    const double status = slot_host[2 * (size_t)rs + (size_t)k * Bx->ed() + 1];
    if (status == 1.0) {
        LKCHK(resident_recover(Bx->ctx));
        *redo = 1;
    } else if (status != 0.0) {
        return fail(LK_ERR_HIP, "the single-launch Gram-Schmidt step failed after its first phase (status %g)", status);
    }
    return LK_OK;
}

// h1 | h2 | ||y''||^2 into the three sections at `out` (stride rs) exactly where dgs_device leaves them, + the normalise and the
// device-side stop test of scal_launch when `normalize` (tol_scale: no scaling below it; tol_break: raises *stop_out).
// Kernel choice: dgs_onchip (the panel stays in registers, X read once) when the row tiles of some shape -- 16 / 8 / 4 columns per wave,
// 2 / 4 / 8 tiles per block -- fit one block per CU; else dgs_resident (three walks served from the caches).
int dgs_resident_launch(lk_basis_t Bx, int k, double *y, double *out, int rs, bool normalize, double tol_scale, double tol_break, int *stop_out,
                        int c0) {
    lk_context_t c = Bx->ctx;
    ResidentWs ws;
    LKCHK(resident_ws(c, &ws));
    const bool cp = Bx->dtype == LK_C128;
    const int ED = Bx->ed();
    constexpr int NW = 8;
    const int maxg = c->num_cu < RES_MAX_GRID ? c->num_cu : RES_MAX_GRID;
    struct Shape { int KC, WC, kcw; int64_t ntiles; };
    auto shape = [&](int KC) {
        Shape sh;
        sh.KC = KC;
        int wcn = (k + KC - 1) / KC, WC = 1;
        while (WC < wcn) WC <<= 1;
        sh.WC = WC;
        sh.kcw = (k + WC - 1) / WC;
        const int64_t tile_rows = (int64_t)(NW / WC) * 64 * (cp ? 1 : 2);
        sh.ntiles = (Bx->n + tile_rows - 1) / tile_rows;
        return sh;
    };
    // among the shapes whose tiles fit: the one that leaves the fewest register slots empty (WC * KC - k), the widest on a tie
    // (fewer wave-columns to exchange between)
    int onchip_kc = 0, waste = INT_MAX;
    if (c->resident_onchip)
        for (int KC : {16, 8, 4}) {
            if (k > KC * NW) continue;
            const Shape sh = shape(KC);
            if (sh.ntiles <= (int64_t)res_onchip_tiles(KC) * maxg && sh.WC * KC - k < waste) { onchip_kc = KC; waste = sh.WC * KC - k; }
        }
    const Shape sh = shape(onchip_kc ? onchip_kc : 16);
    int64_t g = sh.ntiles < maxg ? sh.ntiles : maxg;
    if (g < 1) g = 1;
    const int flags = (normalize ? 1 : 0) | (c->resident_rev ? 2 : 0);
    const long long spin = (long long)c->resident_spin_ms * 100000ll;          // wall_clock64 ticks at 100 MHz
    const double bytes = (double)Bx->n * ED * 8.0 * (3.0 * k + 5.0);
    ProfScope ps(c, "dgs_sweep_resident", bytes, c->prof_ext);
    auto go = [&](auto kern) {
        if (ps.on && ps.ext)
            hipExtLaunchKernelGGL(kern, dim3((unsigned)g), dim3(NW * 64), 0, c->stream, ps.rec.e0, ps.rec.e1, 0, Bx->col(c0), Bx->ld, k, y, Bx->n, ws,
                                  out, rs, sh.WC, sh.kcw, flags, tol_scale, tol_break, stop_out, spin, c->guard());
        else
            hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(NW * 64), 0, c->stream, Bx->col(c0), Bx->ld, k, y, Bx->n, ws, out, rs, sh.WC, sh.kcw,
                               flags, tol_scale, tol_break, stop_out, spin, c->guard());
    };
    switch (onchip_kc) {
    case 16: if (cp) go(dgs_onchip<true, 16, NW>); else go(dgs_onchip<false, 16, NW>); break;
    case 8: if (cp) go(dgs_onchip<true, 8, NW>); else go(dgs_onchip<false, 8, NW>); break;
    case 4: if (cp) go(dgs_onchip<true, 4, NW>); else go(dgs_onchip<false, 4, NW>); break;
    default: if (cp) go(dgs_resident<true, 16, NW>); else go(dgs_resident<false, 16, NW>);
    }
    HIPCHK(hipGetLastError());
    c->resident_stats[0] += 1;
    if (onchip_kc) c->resident_stats[2] += 1;
    return LK_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int lk_version(void) { return 100; }
const char *lk_last_error(void) { return g_err; }

int lk_init(int device, void *stream, lk_context_t *ctx) {
    if (!ctx) return fail(LK_ERR_INVALID, "lk_init: null ctx pointer");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(LK_ERR_HIP, "lk_init: no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(LK_ERR_INVALID, "lk_init: device %d out of range [0,%d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    lk_context_t c = new lk_context_s();
    c->device = device;
    // any failure below releases what was acquired so far (lk_finalize tolerates a half-built context)
    auto body = [&]() -> int {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (stream) {
            c->stream = (hipStream_t)stream;
        } else {
            HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
        HIPCHK(hipMalloc((void **)&c->partial, (size_t)PARTIAL_SECTIONS * RED_SECTION * MAX_GRID * sizeof(double)));
        // sized for the wide layouts too: 9 sections of RED_SECTION_WIDE (the narrow layouts use the head of the same buffer)
        HIPCHK(hipMalloc((void **)&c->red, (size_t)RED_TOTAL * RED_SECTION_WIDE * sizeof(double)));
        HIPCHK(hipMemsetAsync(c->red, 0, (size_t)RED_TOTAL * RED_SECTION_WIDE * sizeof(double), c->stream));
        HIPCHK(hipHostMalloc((void **)&c->red_host, (size_t)RED_TOTAL * RED_SECTION_WIDE * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&c->coef, (size_t)KMAX_WIDE * 2 * sizeof(double)));
        HIPCHK(hipHostMalloc((void **)&c->coef_host, (size_t)KMAX_WIDE * 2 * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&c->stop_dev, sizeof(int)));
        HIPCHK(hipMemsetAsync(c->stop_dev, 0, sizeof(int), c->stream));
        HIPCHK(hipHostMalloc((void **)&c->stop_host, sizeof(int), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&c->coef_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->coef_ev, c->stream));
        return LK_OK;
    };
    const int rc_init = body();
    if (rc_init != LK_OK) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        (void)lk_finalize(c);
        memcpy(g_err, keep, sizeof(keep));
        return rc_init;
    }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        g_live_ctx.insert(c);
    }
    *ctx = c;
    return LK_OK;
}

int lk_context_info(lk_context_t c, int *device, void **stream) {
    if (!c) return fail(LK_ERR_INVALID, "lk_context_info: null context");
    if (device) *device = c->device;
    if (stream) *stream = (void *)c->stream;
    return LK_OK;
}

int lk_comm_info(lk_context_t c, int *nranks, int *rank) {
    if (!c) return fail(LK_ERR_INVALID, "lk_comm_info: null context");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return LK_OK;
}

int lk_finalize(lk_context_t c) {
    if (!c) return LK_OK;
    DevGuard dev_guard(c);
    (void)lk_pool_release_all(c);
    (void)lk_comm_destroy(c);
    if (c->stream) (void)lazy_flush(c);
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        g_live_ctx.erase(c);
    }

    if (c->stream) (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->partial) (void)hipFree(c->partial);
    if (c->red) (void)hipFree(c->red);
    if (c->coef) (void)hipFree(c->coef);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->xhy) (void)hipFree(c->xhy);
    if (c->lz_red) (void)hipFree(c->lz_red);
    if (c->lz_red_host) (void)hipHostFree(c->lz_red_host);
    if (c->red_host) (void)hipHostFree(c->red_host);
    if (c->coef_host) (void)hipHostFree(c->coef_host);
    if (c->coef_ev) (void)hipEventDestroy(c->coef_ev);
    if (c->stop_dev) (void)hipFree(c->stop_dev);
    if (c->stop_host) (void)hipHostFree(c->stop_host);
    if (c->seg_stop_host) (void)hipHostFree(c->seg_stop_host);
    for (auto e : c->seg_events) (void)hipEventDestroy(e);
    c->seg_events.clear();
    if (c->step_red) (void)hipFree(c->step_red);
    if (c->step_red_host) (void)hipHostFree(c->step_red_host);
    if (c->res_cnt) (void)hipFree(c->res_cnt);
    if (c->res_tim) (void)hipFree(c->res_tim);
    if (c->blk_red) (void)hipFree(c->blk_red);
    if (c->blk_red_host) (void)hipHostFree(c->blk_red_host);
    if (c->res_gran) (void)hipFree(c->res_gran);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return LK_OK;
}

int lk_sync(lk_context_t c) {
    if (!c) return fail(LK_ERR_INVALID, "lk_sync: null context");
    DevGuard dev_guard(c);
    LKCHK(lazy_enter(c, false));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    return LK_OK;
}

int lk_set_allreduce(lk_context_t c, lk_allreduce_fn fn, void *user, int nranks, int rank) {
    if (!c) return fail(LK_ERR_INVALID, "lk_set_allreduce: null context");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(LK_ERR_INVALID, "lk_set_allreduce: bad rank %d/%d", rank, nranks);
    if (nranks > 1 && !fn) return fail(LK_ERR_INVALID, "lk_set_allreduce: nranks>1 needs a callback");
    c->allreduce = fn;
    c->allreduce_user = user;
    c->nranks = nranks;
    c->rank = rank;
    return LK_OK;
}

int lk_set_partition(lk_context_t c, int64_t row0, int64_t n_global) {
    if (!c) return fail(LK_ERR_INVALID, "lk_set_partition: null context");
    if (row0 < 0 || (n_global >= 0 && row0 > n_global)) return fail(LK_ERR_INVALID, "lk_set_partition: bad row0");
    c->row0 = row0;
    c->n_global = n_global;
    return LK_OK;
}

int lk_set_tuning(lk_context_t c, const char *key, int value) {
    if (!c || !key) return fail(LK_ERR_INVALID, "lk_set_tuning: null argument");
    if (!strcmp(key, "grid_mult")) {
        if (value < 1 || value > 16) return fail(LK_ERR_INVALID, "grid_mult must be in [1,16]");
        c->grid_mult = value;
        return LK_OK;
    }
    if (!strcmp(key, "dot_colwise")) { c->dot_colwise = value != 0; return LK_OK; }
    if (!strcmp(key, "cw_u")) { c->cw_u = value == 8 ? 8 : (value == 4 ? 4 : 0); return LK_OK; }
    if (!strcmp(key, "cw_grid_mult")) { if (value < 1 || value > 16) return fail(LK_ERR_INVALID, "lk_set_tuning: cw_grid_mult must be in [1, 16]"); c->cw_grid_mult = value; return LK_OK; }
    if (!strcmp(key, "xhy_db")) { c->xhy_db = value < 0 ? 0 : (value > 2 ? 2 : value); return LK_OK; }
    if (!strcmp(key, "xhy_mfma")) { c->xhy_mfma = value != 0; return LK_OK; }
    if (!strcmp(key, "block_fused")) { c->block_fused = value < 0 ? 0 : (value > 2 ? 2 : value); return LK_OK; }
    if (!strcmp(key, "csr_stream")) { c->csr_stream = value != 0; return LK_OK; }
    if (!strcmp(key, "blas1_grid_mult")) {
        if (value < 1 || value > 64) return fail(LK_ERR_INVALID, "lk_set_tuning: blas1_grid_mult must be in [1, 64]");
        c->blas1_grid_mult = value;
        return LK_OK;
    }
    if (!strcmp(key, "resident")) { c->resident = value != 0; if (value) { c->resident_off = false; c->resident_pause = 16; } return LK_OK; }
    if (!strcmp(key, "resident_max_mb")) { c->resident_max_mb = value < 0 ? 0 : value; return LK_OK; }
    if (!strcmp(key, "resident_onchip")) { c->resident_onchip = value != 0; return LK_OK; }
    if (!strcmp(key, "resident_rev")) { c->resident_rev = value != 0; return LK_OK; }
    if (!strcmp(key, "resident_spin_ms")) { c->resident_spin_ms = value < 0 ? 0 : (value > 20000 ? 20000 : value); return LK_OK; }
    if (!strcmp(key, "lazy")) {
        LKCHK(lazy_flush(c));
        c->forget_memos();
        c->lazy = value != 0;
        return LK_OK;
    }
    if (!strcmp(key, "lazy_speculate")) { c->lazy_speculate = value != 0; c->spec.armed = false; return LK_OK; }
    if (!strcmp(key, "recompute_update")) { c->recompute_update = value != 0; return LK_OK; }
    if (!strcmp(key, "store_policy")) {
        if (value < 0 || value > 3) return fail(LK_ERR_INVALID, "store_policy must be in [0,3]");
        c->store_policy = value;
        return LK_OK;
    }
    if (!strcmp(key, "store_split")) { c->store_split = value != 0; return LK_OK; }
    if (!strcmp(key, "gemm_3m")) { c->gemm_3m = value ? 1 : 0; return LK_OK; }
    if (!strcmp(key, "upd_rs")) { c->upd_rs = value != 0; return LK_OK; }
    if (!strcmp(key, "gram_rs")) { c->gram_rs = value < 0 ? 0 : (value > 16 ? 16 : value); return LK_OK; }
#ifdef LK_DIAGNOSTICS
    // phase-timing switches that turn parts of a kernel OFF (wrong results): only in a build made with -DLK_DIAGNOSTICS (make diagnostics), never
    // in the library build() produces
    if (!strcmp(key, "upd_debug")) { c->upd_debug = value & 15; return LK_OK; }
    if (!strcmp(key, "xhy_debug")) { c->xhy_debug = value & 3; return LK_OK; }
#endif
    if (!strcmp(key, "gemm_roll")) { c->gemm_roll = value < 0 ? 0 : (value > 2 ? 2 : value); return LK_OK; }
    if (!strcmp(key, "wide_s3")) { c->wide_s3 = value ? 1 : 0; return LK_OK; }
    if (!strcmp(key, "wide_regs")) { c->wide_regs = value < 0 ? 0 : (value > 2 ? 2 : value); return LK_OK; }
    if (!strcmp(key, "async_arnoldi")) { c->async_arnoldi = value != 0; return LK_OK; }
    if (!strcmp(key, "pool_slab_cols")) {
        if (value < 2 || value > 4096) return fail(LK_ERR_INVALID, "pool_slab_cols must be in [2,4096]");
        c->pool_slab_cols = value;
        return LK_OK;
    }
    if (!strcmp(key, "gemm_mfma_min")) { c->gemm_mfma_min = value < 0 ? 0 : value; return LK_OK; }
    return fail(LK_ERR_INVALID, "lk_set_tuning: unknown key '%s'", key);
}

int lk_profile_enable(lk_context_t c, int on) {
    if (!c) return fail(LK_ERR_INVALID, "null context");
    DevGuard dev_guard(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    c->prof = on != 0;
    return LK_OK;
}
int lk_profile_reset(lk_context_t c) {
    if (!c) return fail(LK_ERR_INVALID, "null context");
    DevGuard dev_guard(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    c->prof_acc.clear();
    return LK_OK;
}
int lk_profile_get(lk_context_t c, const char *tag, int64_t *count, double *total_ms, double *total_bytes) {
    if (!c || !tag) return fail(LK_ERR_INVALID, "null argument");
    DevGuard dev_guard(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    prof_collect(c);
    ProfAcc a;
    const size_t len = strlen(tag);
    if (len > 0 && tag[len - 1] == '*') {   // prefix match: "dgs_sweep*" sums the three sweep kinds
        for (auto &kv : c->prof_acc)
            if (kv.first.compare(0, len - 1, tag, len - 1) == 0) { a.count += kv.second.count; a.ms += kv.second.ms; a.bytes += kv.second.bytes; }
    } else {
        auto it = c->prof_acc.find(tag);
        if (it != c->prof_acc.end()) a = it->second;
    }
    if (count) *count = a.count;
    if (total_ms) *total_ms = a.ms;
    if (total_bytes) *total_bytes = a.bytes;
    return LK_OK;
}

int lk_lazy_fusion_stats(lk_context_t c, int64_t *out4) {
    if (!c || !out4) return fail(LK_ERR_INVALID, "lk_lazy_fusion_stats: null argument");
    for (int i = 0; i < 4; ++i) out4[i] = c->fusion_stats[i];
    return LK_OK;
}

int lk_lazy_speculation_stats(lk_context_t c, int64_t *out2) {
    if (!c || !out2) return fail(LK_ERR_INVALID, "lk_lazy_speculation_stats: null argument");
    out2[0] = c->spec_stats[0];
    out2[1] = c->spec_stats[1];
    return LK_OK;
}

int lk_resident_stats(lk_context_t c, int64_t *out3) {
    if (!c || !out3) return fail(LK_ERR_INVALID, "lk_resident_stats: null argument");
    for (int i = 0; i < 3; ++i) out3[i] = c->resident_stats[i];
    return LK_OK;
}

int lk_resident_phase_ticks(lk_context_t c, int64_t *out8) {
    if (!c || !out8) return fail(LK_ERR_INVALID, "lk_resident_phase_ticks: null argument");
    DevGuard dev_guard(c);
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!c->res_tim) return LK_OK;
    long long t[8];
    HIPCHK(hipMemcpyAsync(t, c->res_tim, sizeof(t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 8; ++i) out8[i] = (int64_t)t[i];
    return LK_OK;
}

int lk_lazy_stats(lk_context_t c, int64_t *out4) {
    if (!c || !out4) return fail(LK_ERR_INVALID, "lk_lazy_stats: null argument");
    for (int i = 0; i < 4; ++i) out4[i] = c->lazy_stats[i];
    return LK_OK;
}

// ---- basis --------------------------------------------------------------------------------
int lk_basis_create(lk_context_t c, int dtype, int64_t n_local, int ncols, lk_basis_t *B) {
    if (!c || !B) return fail(LK_ERR_INVALID, "lk_basis_create: null argument");
    DevGuard dev_guard(c);
    if (dtype != LK_F64 && dtype != LK_C128) return fail(LK_ERR_INVALID, "lk_basis_create: bad dtype %d", dtype);
    if (n_local < 0 || ncols < 1) return fail(LK_ERR_INVALID, "lk_basis_create: bad shape %lld x %d", (long long)n_local, ncols);
    if (n_local > 2147483647LL) return fail(LK_ERR_INVALID, "lk_basis_create: n_local exceeds get_size's default integer");
    const int ed = dtype == LK_C128 ? 2 : 1;
    const int64_t align_elems = 256 / (8 * ed);  // 256-byte column alignment
    int64_t ld = ((n_local + align_elems - 1) / align_elems) * align_elems;
    if (ld == 0) ld = align_elems;
    lk_basis_t b = new lk_basis_s();
    b->ctx = c; b->dtype = dtype; b->n = n_local; b->ld = ld; b->ncols = ncols; b->own = true; b->data = nullptr;
    const size_t bytes = (size_t)ld * ncols * ed * sizeof(double);
    hipError_t e = hipMalloc((void **)&b->data, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // clear the runtime's sticky last-error state: the NEXT launch check must not report this
        delete b;
        return fail(LK_ERR_NOMEM, "lk_basis_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    e = hipMemsetAsync(b->data, 0, bytes, c->stream);
    if (e != hipSuccess) { (void)hipFree(b->data); delete b; return fail(LK_ERR_HIP, "memset failed: %s", hipGetErrorString(e)); }
    *B = b;
    return LK_OK;
}

int lk_basis_wrap(lk_context_t c, int dtype, int64_t n_local, int ncols, int64_t ld, void *dev_ptr, lk_basis_t *B) {
    if (!c || !B || !dev_ptr) return fail(LK_ERR_INVALID, "lk_basis_wrap: null argument");
    if (dtype != LK_F64 && dtype != LK_C128) return fail(LK_ERR_INVALID, "lk_basis_wrap: bad dtype %d", dtype);
    if (n_local < 0 || ncols < 1 || ld < n_local) return fail(LK_ERR_INVALID, "lk_basis_wrap: bad shape");
    if (((uintptr_t)dev_ptr & 15) != 0) return fail(LK_ERR_INVALID, "lk_basis_wrap: pointer must be 16-byte aligned");
    if (dtype == LK_F64 && (ld & 1) && ncols > 1) return fail(LK_ERR_INVALID, "lk_basis_wrap: ld must be even for LK_F64");
    lk_basis_t b = new lk_basis_s();
    b->ctx = c; b->dtype = dtype; b->n = n_local; b->ld = ld; b->ncols = ncols; b->own = false; b->data = (double *)dev_ptr;
    b->hwm = ncols;   // caller-owned memory: every column may hold data
    *B = b;
    return LK_OK;
}

int lk_basis_destroy(lk_basis_t B) {
    if (!B) return LK_OK;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        if (g_live_ctx.count(B->ctx)) {          // a queued update may still target / read this panel
            lk_context_t c = B->ctx;
            DevGuard dev_guard(c);               // the flush below launches on c->stream: run it on the context's device
            auto &q = c->queue;
            // (a non-owning handle is a VIEW of memory that lives on: what is pending there must be written)
            const bool only_target = B->own && q.active && q.zeroed && q.By == B && q.Bx != B && !(c->sub.active && c->sub.By == B);
            if (only_target) q.By = nullptr;     // a virtual temporary dies unwritten; its coefficients may still serve `sub`
            else (void)lazy_flush(c);
            c->forget_memos();
        }
    }
    // hipFree waits for outstanding device work itself; the context may already be finalized.
    if (B->own && B->data) (void)hipFree(B->data);
    delete B;
    return LK_OK;
}

int lk_basis_info(lk_basis_t B, int *dtype, int64_t *n_local, int *ncols, int64_t *ld, void **dev_ptr) {
    if (!B) return fail(LK_ERR_INVALID, "lk_basis_info: null basis");
    if (dtype) *dtype = B->dtype;
    if (n_local) *n_local = B->n;
    if (ncols) *ncols = B->ncols;
    if (ld) *ld = B->ld;
    if (dev_ptr) *dev_ptr = B->data;
    return LK_OK;
}

int lk_vec_device_ptr(lk_basis_t B, int j, int access, void **dev_ptr) {
    LKCHK(check_vec(B, j, "lk_vec_device_ptr"));
    if (!dev_ptr) return fail(LK_ERR_INVALID, "lk_vec_device_ptr: null dev_ptr");
    if (access < LK_ACCESS_READ || access > LK_ACCESS_READWRITE) return fail(LK_ERR_INVALID, "lk_vec_device_ptr: bad access %d", access);
    DevGuard dev_guard(B->ctx);
    const VecRef v{B, j};
    // the caller's kernel sees (and, when writing, defines) the vector's contents: pending work on it is applied first
    LKCHK(lazy_enter_vec(B->ctx, access == LK_ACCESS_READ ? nullptr : &v, access == LK_ACCESS_OVERWRITE,
                         access == LK_ACCESS_OVERWRITE ? nullptr : &v, nullptr));
    if (access != LK_ACCESS_READ) B->touch(j);
    *dev_ptr = B->col(j);
    return LK_OK;
}

int lk_basis_upload(lk_basis_t B, int col0, int ncols, const void *host, int64_t ldh) {
    if (!B || !host) return fail(LK_ERR_INVALID, "lk_basis_upload: null argument");
    DevGuard dev_guard(B->ctx);
    if (col0 < 0 || ncols < 0 || col0 + ncols > B->ncols || ldh < B->n) return fail(LK_ERR_INVALID, "lk_basis_upload: bad range");
    if (ncols == 0 || B->n == 0) return LK_OK;
    if (ncols == 1) {                                               // one vector (y%upload(x)): like any other overwrite of it
        const VecRef w{B, col0};
        LKCHK(lazy_enter_vec(B->ctx, &w, true, nullptr, nullptr));
    } else {
        LKCHK(lazy_enter(B->ctx, true));
    }
    B->touch(col0, ncols);
    const size_t es = (size_t)B->ed() * sizeof(double);
    HIPCHK(hipMemcpy2DAsync(B->col(col0), (size_t)B->ld * es, host, (size_t)ldh * es, (size_t)B->n * es, ncols,
                            hipMemcpyHostToDevice, B->ctx->stream));
    HIPCHK(hipStreamSynchronize(B->ctx->stream));
    return LK_OK;
}

int lk_basis_download(lk_basis_t B, int col0, int ncols, void *host, int64_t ldh) {
    if (!B || !host) return fail(LK_ERR_INVALID, "lk_basis_download: null argument");
    DevGuard dev_guard(B->ctx);
    if (col0 < 0 || ncols < 0 || col0 + ncols > B->ncols || ldh < B->n) return fail(LK_ERR_INVALID, "lk_basis_download: bad range");
    if (ncols == 0 || B->n == 0) return LK_OK;
    if (ncols == 1) {
        const VecRef r{B, col0};
        LKCHK(lazy_enter_vec(B->ctx, nullptr, false, &r, nullptr));
    } else {
        LKCHK(lazy_enter(B->ctx, false));
    }
    const size_t es = (size_t)B->ed() * sizeof(double);
    HIPCHK(hipMemcpy2DAsync(host, (size_t)ldh * es, B->col(col0), (size_t)B->ld * es, (size_t)B->n * es, ncols,
                            hipMemcpyDeviceToHost, B->ctx->stream));
    HIPCHK(hipStreamSynchronize(B->ctx->stream));
    return LK_OK;
}

// ---- column pool ---------------------------------------------------------------------------
static int pool_find_slab(lk_context_t c, lk_basis_t slab) {
    for (size_t i = 0; i < c->pool.size(); ++i)
        if (c->pool[i].B == slab) return (int)i;
    return -1;
}

int lk_pool_acquire(lk_context_t c, int dtype, int64_t n_local, uint64_t tag, lk_basis_t *slab, int *col) {
    if (!c || !slab || !col) return fail(LK_ERR_INVALID, "lk_pool_acquire: null argument");
    if (tag == 0) return fail(LK_ERR_INVALID, "lk_pool_acquire: owner tag 0 is reserved for 'free'");
    DevGuard dev_guard(c);
    auto matches = [&](int si) { return c->pool[si].B->dtype == dtype && c->pool[si].B->n == n_local; };
    auto it = c->pool_by_tag.find(tag);
    if (it != c->pool_by_tag.end()) {
        const int si = it->second.first, cj = it->second.second;
        if (matches(si)) {                                   // the object that lived at this address is gone: re-use
            *slab = c->pool[si].B; *col = cj;
            c->pool[si].gen[cj] = ++c->pool_epoch;           // bit copies of the previous occupant's handle are stale from here on
            c->pool_stats[1] += 1;
            return LK_OK;
        }
        c->pool[si].owner[cj] = 0;                            // other shape: give the old column back
        c->pool_free.insert({si, cj});
        c->pool_by_tag.erase(it);
    }
    for (auto f = c->pool_free.begin(); f != c->pool_free.end(); ++f) {
        if (!matches(f->first)) continue;
        const int si = f->first, cj = f->second;
        c->pool_free.erase(f);
        c->pool[si].owner[cj] = tag;
        c->pool[si].gen[cj] = ++c->pool_epoch;
        c->pool_by_tag[tag] = {si, cj};
        *slab = c->pool[si].B; *col = cj;
        c->pool_stats[1] += 1;
        return LK_OK;
    }
    int si = -1;
    for (int i = (int)c->pool.size() - 1; i >= 0; --i)
        if (matches(i) && c->pool[i].used < c->pool[i].B->ncols) { si = i; break; }
    if (si < 0) {
        // new slab: pool_slab_cols columns.  A single-rank context takes fewer when that would exceed a quarter of the free
        // memory.  A row-sharded context (nranks > 1) NEVER derives the geometry from its own free memory: that differs from
        // rank to rank, V(k) would fall into a second slab at a different k on different ranks, and the lazy path's batched
        // sweeps (column counts from the slab's written columns) would issue all-reduces of different lengths.  There the
        // slab has exactly pool_slab_cols columns on every rank, and an allocation that does not fit fails loudly.
        int ncols = c->pool_slab_cols;
        size_t free_b = 0, total_b = 0;
        if (c->nranks == 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double colbytes = (double)(n_local > 0 ? n_local : 1) * (dtype == LK_C128 ? 16.0 : 8.0);
            const double fit = 0.25 * (double)free_b / colbytes;
            if (fit < ncols) ncols = fit < 8.0 ? 8 : (int)fit;
        }
        lk_context_s::PoolSlab ps;
        {
            const int rc_slab = lk_basis_create(c, dtype, n_local, ncols, &ps.B);
            if (rc_slab != LK_OK && c->nranks > 1) {
                char keep[sizeof(g_err)];
                memcpy(keep, g_err, sizeof(keep));
                return fail(rc_slab, "lk_pool_acquire: a slab of pool_slab_cols = %d columns of %lld rows does not fit on rank %d (%.400s); "
                            "set the SAME smaller pool_slab_cols on every rank (lk_set_tuning)", ncols, (long long)n_local, c->rank, keep);
            }
            LKCHK(rc_slab);
        }
        ps.owner.assign((size_t)ncols, 0);
        ps.gen.assign((size_t)ncols, 0);
        c->pool.push_back(ps);
        si = (int)c->pool.size() - 1;
    }
    auto &ps = c->pool[si];
    const int cj = ps.used++;
    ps.owner[cj] = tag;
    ps.gen[cj] = ++c->pool_epoch;
    c->pool_by_tag[tag] = {si, cj};
    c->pool_stats[0] += 1;
    *slab = ps.B; *col = cj;
    return LK_OK;
}

int lk_pool_owner(lk_context_t c, lk_basis_t slab, int col, uint64_t *tag) {
    if (!c || !tag) return fail(LK_ERR_INVALID, "lk_pool_owner: null argument");
    *tag = 0;
    const int si = pool_find_slab(c, slab);
    if (si < 0 || col < 0 || col >= c->pool[si].used) return LK_OK;
    *tag = c->pool[si].owner[col];
    return LK_OK;
}

int lk_pool_column_info(lk_context_t c, lk_basis_t slab, int col, uint64_t *tag, uint64_t *generation) {
    if (!c) return fail(LK_ERR_INVALID, "lk_pool_column_info: null context");
    if (tag) *tag = 0;
    if (generation) *generation = 0;
    const int si = pool_find_slab(c, slab);
    if (si < 0 || col < 0 || col >= c->pool[si].used) return LK_OK;
    if (tag) *tag = c->pool[si].owner[col];
    if (generation) *generation = c->pool[si].gen[col];
    return LK_OK;
}

int lk_pool_release(lk_context_t c, lk_basis_t slab, int col) {
    if (!c) return fail(LK_ERR_INVALID, "lk_pool_release: null context");
    DevGuard dev_guard(c);
    const int si = pool_find_slab(c, slab);
    if (si < 0 || col < 0 || col >= c->pool[si].used) return fail(LK_ERR_INVALID, "lk_pool_release: not a pool column");
    const uint64_t tag = c->pool[si].owner[col];
    if (tag == 0) return LK_OK;
    if (c->lazy && c->queue.active && c->queue.By == slab && c->queue.jy == col) {   // the column's contents die with its owner
        if (c->queue.zeroed) c->queue.By = nullptr;
        else c->queue.active = false;
    }
    c->pool_by_tag.erase(tag);
    c->pool[si].owner[col] = 0;
    c->pool_free.insert({si, col});
    return LK_OK;
}

int lk_pool_release_all(lk_context_t c) {
    if (!c) return LK_OK;
    DevGuard dev_guard(c);
    for (auto &ps : c->pool) (void)lk_basis_destroy(ps.B);
    c->pool.clear();
    c->pool_by_tag.clear();
    c->pool_free.clear();
    return LK_OK;
}

int lk_pool_stats(lk_context_t c, int64_t *out4) {
    if (!c || !out4) return fail(LK_ERR_INVALID, "lk_pool_stats: null argument");
    out4[0] = (int64_t)c->pool.size();
    out4[1] = c->pool_stats[0];
    out4[2] = (int64_t)c->pool_by_tag.size();
    out4[3] = c->pool_stats[1];
    return LK_OK;
}

// ---- vector TBPs ----------------------------------------------------------------------------
int lk_vec_zero(lk_basis_t B, int j) {
    LKCHK(check_vec(B, j, "lk_vec_zero"));
    DevGuard dev_guard(B->ctx);
    lk_context_t c = B->ctx;
    const VecRef w{B, j};
    LKCHK(lazy_enter_vec(c, &w, true, nullptr, nullptr));
    B->touch(j);
    if (c->lazy) {
        // proj%zero() opens linear_combination (AbstractVectors.fypp:598): the memset is deferred with the queue
        LKCHK(materialise_queue(c));                             // one slot: an older virtual vector becomes real
        auto &q = c->queue;
        q.active = true; q.zeroed = true; q.By = B; q.jy = j; q.Bx = nullptr; q.j0 = 0; q.cnt = 0; q.coef.clear();
        return LK_OK;
    }
    HIPCHK(hipMemsetAsync(B->col(j), 0, (size_t)B->n * B->ed() * sizeof(double), c->stream));
    return LK_OK;
}

int lk_vec_scal(lk_basis_t B, int j, const double *alpha) {
    LKCHK(check_vec(B, j, "lk_vec_scal"));
    DevGuard dev_guard(B->ctx);
    if (!alpha) return fail(LK_ERR_INVALID, "lk_vec_scal: null alpha");
    const VecRef w{B, j};
    LKCHK(lazy_enter_vec(B->ctx, &w, false, &w, nullptr));
    B->touch(j);
    return scal_launch(B, j, alpha[0], B->dtype == LK_C128 ? alpha[1] : 0.0, nullptr, 0.0);
}

int lk_vec_axpby(const double *alpha, lk_basis_t Bx, int jx, const double *beta, lk_basis_t By, int jy) {
    LKCHK(check_vec(Bx, jx, "lk_vec_axpby(vec)"));
    LKCHK(check_vec(By, jy, "lk_vec_axpby(self)"));
    DevGuard dev_guard(Bx->ctx);
    LKCHK(check_pair(Bx, By, "lk_vec_axpby"));
    if (!alpha || !beta) return fail(LK_ERR_INVALID, "lk_vec_axpby: null scalar");
    lk_context_t c = Bx->ctx;
    const bool cp = Bx->dtype == LK_C128;
    By->touch(jy);
    if (c->lazy) {
        const int ED = Bx->ed();
        const bool unit_beta = beta[0] == 1.0 && (!cp || beta[1] == 0.0);
        const bool zero_beta = beta[0] == 0.0 && (!cp || beta[1] == 0.0);
        const double *xp = Bx->col(jx);
        double *yp = By->col(jy);
        auto &q = c->queue;
        const double *T = (q.active && q.By) ? q.By->col(q.jy) : nullptr;
        auto in_xrange = [&](const double *p) {
            if (!q.active || q.cnt == 0) return false;
            const double *lo = q.Bx->col(q.j0), *hi = lo + (int64_t)q.cnt * q.Bx->ld * q.Bx->ed();
            return p >= lo && p < hi;
        };
        // y%sub(proj) / x%add(dx) with the operand still virtual (gram_schmidt.fypp:145, gmres.fypp:202): y += s X a pending
        if (unit_beta && q.active && q.zeroed && q.cnt >= 1 && !c->sub.active && T && xp == T && yp != T && !in_xrange(yp) &&
            q.Bx->ctx == By->ctx && q.Bx->n == By->n) {
            c->forget_memos();
            c->sub.active = true; c->sub.By = By; c->sub.jy = jy;
            c->sub.s[0] = alpha[0]; c->sub.s[1] = cp ? alpha[1] : 0.0;
            return LK_OK;
        }
        // y <- a X(:, jx) + 1 y with X a column of a multi-column panel: queue it (linear_combination's loop,
        // AbstractVectors.fypp:600-602 / 637-642); consecutive columns onto the same y extend the queue.
        if (unit_beta && Bx->ncols > 1 && xp != yp && xp != T) {
            LKCHK(apply_sub(c));                                 // an earlier y update used the queue as it was
            const bool same_target = q.active && q.By && q.By->col(q.jy) == yp;
            const bool extends = same_target && q.cnt < KMAX_WIDE &&
                                 (q.cnt == 0 || (q.Bx == Bx && jx == q.j0 + q.cnt));
            if (!extends) {
                const VecRef w{By, jy}, r{Bx, jx};
                LKCHK(lazy_enter_vec(c, &w, false, &r, &w));
                LKCHK(materialise_queue(c));
                q.active = true; q.zeroed = false; q.Bx = Bx; q.By = By; q.jy = jy; q.j0 = jx; q.cnt = 0; q.coef.clear();
            } else if (q.cnt == 0) {
                q.Bx = Bx; q.j0 = jx;
            }
            c->forget_memos();       // y is (about to be) modified
            for (int e = 0; e < ED; ++e) q.coef.push_back(alpha[e]);
            q.cnt += 1;
            c->lazy_stats[2] += 1;
            return LK_OK;
        }
        const VecRef w{By, jy}, r{Bx, jx};
        LKCHK(lazy_enter_vec(c, &w, zero_beta, &r, zero_beta ? nullptr : &w));
    }
    const int64_t nv = Bx->n * Bx->ed() / 2 + 1;
    ProfScope ps(c, "blas1", (double)Bx->n * Bx->ed() * 24.0);
    if (cp)
        hipLaunchKernelGGL(k_axpby<true>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, alpha[0], alpha[1], Bx->col(jx),
                           beta[0], beta[1], By->col(jy), Bx->n, blas1_nt(Bx));
    else
        hipLaunchKernelGGL(k_axpby<false>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, alpha[0], 0.0, Bx->col(jx),
                           beta[0], 0.0, By->col(jy), Bx->n, blas1_nt(Bx));
    HIPCHK(hipGetLastError());
    return LK_OK;
}

// see lk_context_s::spec.  Returns 1 when the norm of column j was served by an anticipated first-pass sweep (memos filled).
static int speculative_first_pass(lk_context_t c, lk_basis_t B, int j, int *done) {
    *done = 0;
    auto &sp = c->spec;
    if (!c->lazy || !c->lazy_speculate || !sp.armed || sp.xbase != B->data || j != sp.jy + 1) return LK_OK;
    if (sp.unused) { sp.armed = false; c->spec_stats[1] += 1; return LK_OK; }      // the last prediction was wasted: stop predicting
    const int cnt = j - sp.j0;
    if (cnt < 2 || cnt > KMAX_WIDE || j >= B->ncols || B->hwm <= j || c->sub.active) return LK_OK;
    // (a virtual temporary elsewhere -- the previous step's last projection, never asked for -- does not matter; one INSIDE the
    //  swept columns does: its contents are not in memory)
    if (c->queue.active && c->queue.By && c->queue.By->data == B->data && c->queue.jy >= sp.j0 && c->queue.jy <= j) return LK_OK;
    const int ED = B->ed();
    double *y = B->col(j);
    LKCHK((sweepm<1>(B, sp.j0, cnt, y, nullptr, nullptr, 0, c->red)));
    LKCHK(fetch(c, 0, 1, red_stride(cnt)));
    auto &mm = c->memo;
    mm.vals.assign(c->red_host, c->red_host + (size_t)cnt * ED);
    mm.valid = true; mm.xbase = B->data; mm.y = y; mm.j0 = sp.j0; mm.cnt = cnt;
    c->nmemo.valid = true; c->nmemo.y = y; c->nmemo.nrm2 = c->red_host[(size_t)cnt * ED];
    sp.jy = j; sp.unused = true;
    c->spec_stats[0] += 1;
    c->lazy_stats[1] += 1;
    *done = 1;
    return LK_OK;
}

int lk_vec_dot(lk_basis_t Bx, int jx, lk_basis_t By, int jy, double *out) {
    LKCHK(check_vec(Bx, jx, "lk_vec_dot(self)"));
    LKCHK(check_vec(By, jy, "lk_vec_dot(vec)"));
    DevGuard dev_guard(Bx->ctx);
    LKCHK(check_pair(Bx, By, "lk_vec_dot"));
    if (!out) return fail(LK_ERR_INVALID, "lk_vec_dot: null out");
    lk_context_t c = Bx->ctx;
    // y%norm() reaches the plugin as y%dot(y): LightKrylov's norm is sqrt(abs(self%dot(self))) (AbstractVectors.fypp:424-432)
    const bool self_dot = Bx->col(jx) == By->col(jy);
    if (c->lazy && c->sub.active && c->sub.By->col(c->sub.jy) == By->col(jy) &&
        (self_dot || (Bx->data == c->queue.Bx->data && jx >= c->queue.j0 && jx < c->queue.j0 + c->queue.cnt))) {
        LKCHK(fused_sub_with_dots(c));                           // the norm / the first dot of the next pass: one fused sweep
    } else {
        const VecRef r1{Bx, jx}, r2{By, jy};
        LKCHK(lazy_enter_vec(c, nullptr, false, &r1, &r2));
    }
    if (c->lazy && self_dot && !(c->nmemo.valid && c->nmemo.y == By->col(jy))) {
        int done = 0;
        LKCHK(speculative_first_pass(c, By, jy, &done));       // the norm that opens a Gram-Schmidt pass: the whole first pass at once
    }
    if (c->lazy && self_dot && c->nmemo.valid && c->nmemo.y == By->col(jy)) {
        out[0] = c->nmemo.nrm2;
        if (Bx->dtype == LK_C128) out[1] = 0.0;
        c->lazy_stats[0] += 1;
        return LK_OK;
    }
    if (c->lazy) {
        if (self_dot) c->spec.last_norm_y = By->col(jy);       // a plain norm: remember it (the dots that follow may complete the pattern)
        else if (c->spec.last_norm_y != By->col(jy)) c->spec.last_norm_y = nullptr;
    }
    if (c->lazy && c->xhy_mfma && Bx->data == By->data && Bx->ncols > 1) {
        // gram_matrix's loop: see gmemo
        auto &gm = c->gmemo;
        const int ED = Bx->ed();
        auto serve = [&]() {
            for (int e = 0; e < ED; ++e) out[e] = gm.G[((size_t)jy * gm.cnt + jx) * ED + e];
            c->lazy_stats[0] += 1;
        };
        if (gm.valid && gm.xbase == Bx->data && jx < gm.cnt && jy < gm.cnt) { serve(); return LK_OK; }
        const bool run = gm.last_x == Bx->data && gm.last_jx == jx && gm.last_jy + 1 == jy;
        gm.last_x = Bx->data; gm.last_jx = jx; gm.last_jy = jy;
        int cnt = Bx->hwm < Bx->ncols ? Bx->hwm : Bx->ncols;
        if (cnt > XHY_MAX) cnt = XHY_MAX;
        if (run && !c->queue.active && !c->sub.active && cnt >= XHY_MIN_P && jx < cnt && jy < cnt) {
            double *dev = nullptr;
            LKCHK(dots_mfma(Bx, 0, cnt, Bx, 0, cnt, 1 | 2, 0, &dev));
            std::vector<double> host((size_t)cnt * (cnt + 1) * ED);
            HIPCHK(hipMemcpyAsync(host.data(), dev, host.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (c->prof) prof_collect(c);
            gm.G.assign((size_t)cnt * cnt * ED, 0.0);
            for (int j = 0; j < cnt; ++j)
                for (int i = 0; i <= j; ++i)
                    for (int e = 0; e < ED; ++e) {
                        const double v = host[((size_t)j * (cnt + 1) + i) * ED + e];
                        gm.G[((size_t)j * cnt + i) * ED + e] = v;
                        gm.G[((size_t)i * cnt + j) * ED + e] = e ? -v : v;        // conj(X_j) . X_i = conj(conj(X_i) . X_j)
                    }
            if (ED == 2)
                for (int j = 0; j < cnt; ++j) gm.G[((size_t)j * cnt + j) * 2 + 1] = 0.0;   // x . x is real (dotc sums exact zeros)
            gm.valid = true; gm.xbase = Bx->data; gm.cnt = cnt;
            c->lazy_stats[1] += 1;
            serve();
            return LK_OK;
        }
    }
    if (c->lazy && Bx->ncols > 1) {
        // innerprod's loop (AbstractVectors.fypp:672-674, 690-694) asks X(1)%dot(y), X(2)%dot(y), ...:
        // the first miss computes the dots of the whole run of columns in ONE sweep, the rest are memo hits.
        const int ED = Bx->ed();
        auto &mm = c->memo;
        const double *yp = By->col(jy);
        if (mm.valid && mm.xbase == Bx->data && mm.y == yp && jx >= mm.j0 && jx < mm.j0 + mm.cnt) {
            for (int e = 0; e < ED; ++e) out[e] = mm.vals[(size_t)(jx - mm.j0) * ED + e];
            c->lazy_stats[0] += 1;
            if (c->spec.armed && c->spec.xbase == Bx->data && c->spec.jy == jy) c->spec.unused = false;   // the anticipated sweep was wanted
            return LK_OK;
        }
        // X(:k) vs y = X(k+1) in the same panel; otherwise the run of columns ever written (slab panels are mostly
        // unused columns: sweeping them would cost up to 128/k times the eager traffic)
        int jend = (By->data == Bx->data && jy > jx) ? jy : (Bx->hwm < Bx->ncols ? Bx->hwm : Bx->ncols);
        int cnt = jend - jx;
        if (cnt > KMAX_WIDE) cnt = KMAX_WIDE;               // one (lane-split) sweep holds up to 512 columns
        if (c->queue.active && c->queue.By && c->queue.By->data == Bx->data && c->queue.jy > jx && c->queue.jy < jx + cnt)
            cnt = c->queue.jy - jx;                              // never sweep a column whose contents are still virtual
        const bool y_inside = (By->data == Bx->data) && jy >= jx && jy < jx + cnt;
        if (cnt >= 2 && !y_inside) {
            LKCHK((sweepm<1>(Bx, jx, cnt, By->col(jy), nullptr, nullptr, 0, c->red)));
            LKCHK(fetch(c, 0, 1, red_stride(cnt)));
            mm.vals.assign(c->red_host, c->red_host + (size_t)cnt * ED);
            mm.valid = true; mm.xbase = Bx->data; mm.y = yp; mm.j0 = jx; mm.cnt = cnt;
            c->lazy_stats[1] += 1;
            // norm of column jy, then the dots of ALL columns before it against it: the opening of a Gram-Schmidt pass against
            // X(:jy) with y = X(jy + 1) of the same panel.  The next step's norm may run this sweep at once (spec).
            if (By->data == Bx->data && jx < jy && cnt == jy - jx && c->spec.last_norm_y == yp) {
                c->spec.armed = true; c->spec.xbase = Bx->data; c->spec.jy = jy; c->spec.j0 = jx; c->spec.unused = false;
            }
            for (int e = 0; e < ED; ++e) out[e] = mm.vals[e];
            return LK_OK;
        }
    }
    LKCHK(dot_device(Bx, jx, By, jy, c->red));
    LKCHK(fetch(c, 0, 1));
    out[0] = c->red_host[0];
    if (Bx->dtype == LK_C128) out[1] = c->red_host[1];
    return LK_OK;
}

int lk_vec_norm(lk_basis_t B, int j, double *out) {
    LKCHK(check_vec(B, j, "lk_vec_norm"));
    DevGuard dev_guard(B->ctx);
    if (!out) return fail(LK_ERR_INVALID, "lk_vec_norm: null out");
    lk_context_t c = B->ctx;
    if (c->lazy && c->sub.active && c->sub.By->col(c->sub.jy) == B->col(j)) {
        // the zero-vector check that opens the next Gram-Schmidt pass (gram_schmidt.fypp:126), or qr's norm after the
        // last one (qr.fypp:135): y' = y - X h is formed here, together with X^H y' for the dot calls that follow
        LKCHK(fused_sub_with_dots(c));
    } else {
        const VecRef r{B, j};
        LKCHK(lazy_enter_vec(c, nullptr, false, &r, nullptr));
    }
    if (c->lazy && !(c->nmemo.valid && c->nmemo.y == B->col(j))) {
        int done = 0;
        LKCHK(speculative_first_pass(c, B, j, &done));
    }
    if (c->lazy && c->nmemo.valid && c->nmemo.y == B->col(j)) {
        *out = std::sqrt(std::fabs(c->nmemo.nrm2));
        c->lazy_stats[0] += 1;
        return LK_OK;
    }
    if (c->lazy) c->spec.last_norm_y = B->col(j);
    LKCHK(dot_device(B, j, B, j, c->red));
    LKCHK(fetch(c, 0, 1));
    // alpha = abs(self%dot(self)); alpha = sqrt(alpha)   AbstractVectors.fypp:431
    *out = std::sqrt(std::hypot(c->red_host[0], B->dtype == LK_C128 ? c->red_host[1] : 0.0));
    return LK_OK;
}

int lk_vec_size(lk_basis_t B, int64_t *n_local) {
    if (!B || !n_local) return fail(LK_ERR_INVALID, "lk_vec_size: null argument");
    *n_local = B->n;
    return LK_OK;
}

int lk_vec_copy(lk_basis_t Bd, int jd, lk_basis_t Bs, int js) {
    LKCHK(check_vec(Bd, jd, "lk_vec_copy(out)"));
    LKCHK(check_vec(Bs, js, "lk_vec_copy(from)"));
    DevGuard dev_guard(Bd->ctx);
    LKCHK(check_pair(Bd, Bs, "lk_vec_copy"));
    if (Bd->col(jd) == Bs->col(js)) return LK_OK;
    {
        const VecRef w{Bd, jd}, r{Bs, js};
        LKCHK(lazy_enter_vec(Bd->ctx, &w, true, &r, nullptr));
    }
    Bd->touch(jd);
    ProfScope ps(Bd->ctx, "blas1", (double)Bd->n * Bd->ed() * 16.0);
    const int64_t nd = Bd->n * Bd->ed();
    hipLaunchKernelGGL(k_copy, dim3(blas1_grid(Bd->ctx, nd / 2 + 1)), dim3(256), 0, Bd->ctx->stream, Bs->col(js), Bd->col(jd), nd,
                       blas1_nt(Bd));
    HIPCHK(hipGetLastError());
    return LK_OK;
}

int lk_vec_rand(lk_basis_t B, int j, uint64_t seed, int64_t row0, int ifnorm) {
    LKCHK(check_vec(B, j, "lk_vec_rand"));
    DevGuard dev_guard(B->ctx);
    lk_context_t c = B->ctx;
    {
        const VecRef w{B, j};
        LKCHK(lazy_enter_vec(c, &w, true, nullptr, nullptr));
    }
    B->touch(j);
    if (B->dtype == LK_C128)
        hipLaunchKernelGGL(k_rand<true>, dim3(blas1_grid(c, B->n)), dim3(256), 0, c->stream, B->col(j), B->n, seed, row0);
    else
        hipLaunchKernelGGL(k_rand<false>, dim3(blas1_grid(c, B->n)), dim3(256), 0, c->stream, B->col(j), B->n, seed, row0);
    HIPCHK(hipGetLastError());
    if (ifnorm) {
        LKCHK(dot_device(B, j, B, j, c->red));
        LKCHK(scal_launch(B, j, 1.0, 0.0, c->red, 0.0));
    }
    return LK_OK;
}

// ---- basis helpers ---------------------------------------------------------------------------
// upper_only (lk_gram): only entries M(i, j) with i <= j of a diagonal block are wanted
static int innerprod_impl(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int p, double *M, bool upper_only) {
    if (!Bx || !By || !M) return fail(LK_ERR_INVALID, "lk_innerprod: null argument");
    DevGuard dev_guard(Bx->ctx);
    LKCHK(check_pair(Bx, By, "lk_innerprod"));
    if (k < 1 || k > Bx->ncols || p < 1 || jy0 < 0 || jy0 + p > By->ncols) return fail(LK_ERR_INVALID, "lk_innerprod: bad range");
    lk_context_t c = Bx->ctx;
    LKCHK(lazy_enter(c, false));
    const int ED = Bx->ed();
    // X^H X entries on the diagonal, complex kind: conj(x) . x has an imaginary part of exactly zero in the reference's dotc (every term is
    // re*im - im*re, no contraction across the two products); the three-product matrix-core kernels form Im as P3 + P1 - P2, the doubled real
    // problem and the FMA-contracted vector kernels round one of the two products -- each leaves O(eps |x|^2) there.  Zeroed, whatever kernel ran.
    auto real_diagonal = [&]() {
        if (ED != 2 || Bx->data != By->data) return;
        for (int q = 0; q < p; ++q) {
            const int i = jy0 + q;                     // column i of X is column q of Y
            if (i < k) M[((size_t)q * k + i) * 2 + 1] = 0.0;
        }
    };
    if (c->xhy_mfma && p >= XHY_MIN_P) {
        // many right-hand sides: one pass over X per 128 x 128 block of M on the matrix cores
        std::vector<double> host((size_t)XHY_MAX * (XHY_MAX + 1) * 2);
        for (int j = 0; j < p; j += XHY_MAX) {
            const int pn = (p - j) < XHY_MAX ? (p - j) : XHY_MAX;
            for (int c0 = 0; c0 < k; c0 += XHY_MAX) {
                const int kk = (k - c0) < XHY_MAX ? (k - c0) : XHY_MAX;
                const bool same = Bx->data == By->data && jy0 + j == c0 && pn == kk;          // a diagonal block of X^H X
                double *out = nullptr;
                LKCHK(dots_mfma(Bx, c0, kk, By, jy0 + j, pn, same ? (1 | (upper_only ? 2 : 0)) : 0, 0, &out));
                HIPCHK(hipMemcpyAsync(host.data(), out, (size_t)pn * (kk + 1) * ED * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if (c->prof) prof_collect(c);
                for (int q = 0; q < pn; ++q)
                    memcpy(M + ((size_t)(j + q) * k + c0) * ED, host.data() + (size_t)q * (kk + 1) * ED, (size_t)kk * ED * sizeof(double));
            }
        }
        real_diagonal();
        return LK_OK;
    }
    for (int j = 0; j < p; j += 4) {
        const int pn = (p - j) < 4 ? (p - j) : 4;
        if (pn == 1) {                                 // one vector: up to KMAX_WIDE columns per sweep
            for (int c0 = 0; c0 < k; c0 += KMAX_WIDE) {
                const int kk = (k - c0) < KMAX_WIDE ? (k - c0) : KMAX_WIDE;
                LKCHK((sweepm<1>(Bx, c0, kk, By->col(jy0 + j), nullptr, nullptr, 0, c->red)));
                LKCHK(fetch(c, 0, 1, red_stride(kk)));
                memcpy(M + ((size_t)j * k + c0) * ED, c->red_host, (size_t)kk * ED * sizeof(double));
            }
            continue;
        }
        for (int c0 = 0; c0 < k; c0 += KMAX_FUSED) {
            const int kk = (k - c0) < KMAX_FUSED ? (k - c0) : KMAX_FUSED;
            {                                   // up to four columns of Y per pass over X
                LKCHK(dots_p(Bx, c0, kk, By, jy0 + j, pn));
                LKCHK(fetch(c, 0, RED_MULTI));
                for (int q = 0; q < pn; ++q)
                    memcpy(M + ((size_t)(j + q) * k + c0) * ED, c->red_host + (size_t)q * (kk + 1) * ED,
                           (size_t)kk * ED * sizeof(double));
            }
        }
    }
    real_diagonal();
    return LK_OK;
}

int lk_innerprod(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int p, double *M) { return innerprod_impl(Bx, k, By, jy0, p, M, false); }

int lk_gram(lk_basis_t Bx, int k, double *G) {
    if (!Bx || !G) return fail(LK_ERR_INVALID, "lk_gram: null argument");
    if (k < 1 || k > Bx->ncols) return fail(LK_ERR_INVALID, "lk_gram: bad k");
    const int ED = Bx->ed();
    std::vector<double> M((size_t)k * k * ED);
    LKCHK(innerprod_impl(Bx, k, Bx, 0, k, M.data(), true));
    // G(i,j) = X(i)%dot(X(j)) for j >= i; G(j,i) = G(i,j) (no conjugation)   AbstractVectors.fypp:650-655
    for (int i = 0; i < k; ++i)
        for (int j = i; j < k; ++j)
            for (int e = 0; e < ED; ++e) {
                const double v = M[((size_t)j * k + i) * ED + e];
                G[((size_t)j * k + i) * ED + e] = v;
                G[((size_t)i * k + j) * ED + e] = v;
            }
    return LK_OK;
}

int lk_lincomb(lk_basis_t Bx, int k, const double *C, int q, lk_basis_t By, int jy0) {
    if (!Bx || !By || !C) return fail(LK_ERR_INVALID, "lk_lincomb: null argument");
    DevGuard dev_guard(Bx->ctx);
    LKCHK(check_pair(Bx, By, "lk_lincomb"));
    if (k < 1 || k > Bx->ncols || q < 1 || jy0 < 0 || jy0 + q > By->ncols)
        return fail(LK_ERR_INVALID, "Krylov basis X and combination matrix B have incompatible sizes.");
    lk_context_t c = Bx->ctx;
    LKCHK(lazy_enter(c, true));
    const int ED = Bx->ed();
    By->touch(jy0, q);
    // the output columns must not be among the inputs (the reference writes into a fresh Xwrk / proj)
    if (Bx->data == By->data && jy0 < k) return fail(LK_ERR_INVALID, "lk_lincomb: output columns alias the input basis");
    // coefficients: one upload of the whole k x q block, repacked on the device for the kernel's scalar loads
    const int64_t raw = (int64_t)k * q * ED;
    LKCHK(ensure_scratch(c, raw + gemm_packed_doubles(k, q, ED)));
    HIPCHK(hipMemcpyAsync(c->scratch, C, (size_t)raw * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LKCHK(gemm_launch(Bx, 0, k, By, jy0, q, c->scratch, (int64_t)k, 1.0, 0, c->scratch + raw));
    HIPCHK(hipStreamSynchronize(c->stream));   // the host coefficient array may be released by the caller
    if (c->prof) prof_collect(c);
    return LK_OK;
}

static int dgs_generic(lk_basis_t Bx, int k, lk_basis_t By, int jy, double *h, double *norms, int flags, int *info,
                       bool two_pass) {
    if (!Bx || !By) return fail(LK_ERR_INVALID, "double_gram_schmidt_step: null basis");
    DevGuard dev_guard(Bx->ctx);
    LKCHK(check_vec(By, jy, "double_gram_schmidt_step(y)"));
    LKCHK(check_pair(Bx, By, "double_gram_schmidt_step"));
    if (k < 1 || k > Bx->ncols) return fail(LK_ERR_INVALID, "double_gram_schmidt_step: k=%d out of range [1,%d]", k, Bx->ncols);
    lk_context_t c = Bx->ctx;
    LKCHK(lazy_enter(c, true));
    const int ED = Bx->ed();
    double *y = By->col(jy);
    {   // y must not lie inside the device range of X(:, :k) -- compared by address, so a wrapped view of the same panel counts
        const double *x0 = Bx->col(0), *x1 = Bx->col(k - 1) + Bx->n * ED;
        if (y + Bx->n * ED > x0 && y < x1) return fail(LK_ERR_INVALID, "double_gram_schmidt_step: y is one of the basis columns");
    }
    double n0 = 0, n1 = 0, n2 = 0;
    resident_maybe_rearm(c);
    bool single = two_pass && k <= KMAX_WIDE && resident_applies(Bx, k);
    if (!single && two_pass) resident_note_fallback(Bx, k);
    const double dgs_bytes = (double)Bx->n * ED * 8.0 * (two_pass ? (3.0 * k + 5.0) : (2.0 * k + 3.0));
    // "dgs" = the step on the device.  Three sweeps: stream markers either side of the six kernels.  Single launch: the kernel's own dispatch
    // timestamps (as inside the asynchronous batches) -- two marker packets around ONE 20 us kernel would add half of its time to it.
    ProfScope ps(c, "dgs", dgs_bytes, false, !single);
    if (k <= KMAX_WIDE) {
        const int rs = red_stride(k);
        const int last = two_pass ? 2 : 1;
        if (single) {
            // the whole step -- both passes and the normalise -- as ONE persistent launch (lk_resident.hip.h)
            c->span_first = nullptr;
            LKCHK(dgs_resident_launch(Bx, k, y, c->red, rs, (flags & LK_DGS_NORMALIZE) != 0, ATOL_DP, 0.0, nullptr));
            if (c->prof && c->span_first) {
                ProfRec span;
                span.e0 = c->span_first; span.e1 = c->span_last; span.tag = "dgs"; span.borrowed = true;
                span.bytes = dgs_bytes;
                c->prof_pending.push_back(span);
            }
            LKCHK(fetch(c, 0, 3, rs));
            const double status = c->red_host[2 * rs + (size_t)k * ED + 1];
            if (status == 1.0) {            // gave up before touching y (the chip is shared with another persistent kernel)
                LKCHK(resident_recover(c));
                resident_note_fallback(Bx, k);
                single = false;
            } else if (status != 0.0) {
                return fail(LK_ERR_HIP, "double_gram_schmidt_step: the single-launch step failed after its first phase (status %g)", status);
            }
        }
        if (!single) {
            LKCHK(dgs_device(Bx, k, y, two_pass));
            if (flags & LK_DGS_NORMALIZE)
                LKCHK(scal_launch(By, jy, 1.0, 0.0, c->red + last * rs + (size_t)k * ED, ATOL_DP));
            ps.end();
            LKCHK(fetch(c, 0, 3, rs));
        }
        const double *r0 = c->red_host, *r1 = c->red_host + rs, *r2 = c->red_host + 2 * rs;
        if (h)
            for (int i = 0; i < k * ED; ++i) h[i] = two_pass ? (r0[i] + r1[i]) : r0[i];   // gram_schmidt.fypp:49
        n0 = r0[k * ED];
        n1 = r1[k * ED];
        n2 = two_pass ? r2[k * ED] : n1;
    } else if (k <= 4 * KMAX_WIDE) {
        // very wide basis, up to four column panels of KMAX_WIDE: everything stays on the device, ONE copy + synchronisation.
        //   sweep 1 : h1_p = X_p^H y for every panel                                   (k columns)
        //   pass B  : y' = y - X h1 -- update-only sweeps for all panels but the last, whose sweep is the fused update + dot
        //             (h2 of that panel, ||y'||^2); then dot sweeps of the OTHER panels against the finished y'   (2k - |last| columns)
        //   pass C  : y'' = y' - X h2, panel by panel, ||y''||^2 from the last                      (k columns)
        // = 4k - |last panel| columns against 4k (+ a host round trip per panel) for the schedule below.
        const int npan = (k + KMAX_WIDE - 1) / KMAX_WIDE, last = npan - 1;
        constexpr int RS = RED_SECTION_WIDE;
        auto c0 = [&](int p) { return p * KMAX_WIDE; };
        auto kk = [&](int p) { return (k - c0(p)) < KMAX_WIDE ? (k - c0(p)) : KMAX_WIDE; };
        auto r1 = [&](int p) { return c->red + (size_t)p * RS; };
        auto r2 = [&](int p) { return c->red + (size_t)(npan + p) * RS; };
        double *r3 = c->red + (size_t)(2 * npan) * RS;
        for (int p = 0; p < npan; ++p) LKCHK((sweepm<1>(Bx, c0(p), kk(p), y, nullptr, nullptr, 0, r1(p))));
        if (two_pass) {
            for (int p = 0; p < last; ++p) LKCHK((sweepm<3>(Bx, c0(p), kk(p), y, r1(p), nullptr, 1, nullptr)));
            LKCHK((sweepm<2>(Bx, c0(last), kk(last), y, r1(last), nullptr, 1, r2(last))));
            for (int p = 0; p < last; ++p) LKCHK((sweepm<1>(Bx, c0(p), kk(p), y, nullptr, nullptr, 0, r2(p))));
            for (int p = 0; p < last; ++p) LKCHK((sweepm<3>(Bx, c0(p), kk(p), y, r2(p), nullptr, 1, nullptr)));
            LKCHK((sweepm<3>(Bx, c0(last), kk(last), y, r2(last), nullptr, 1, r3)));
        } else {
            for (int p = 0; p < last; ++p) LKCHK((sweepm<3>(Bx, c0(p), kk(p), y, r1(p), nullptr, 1, nullptr)));
            LKCHK((sweepm<3>(Bx, c0(last), kk(last), y, r1(last), nullptr, 1, r3)));
        }
        if (flags & LK_DGS_NORMALIZE) LKCHK(scal_launch(By, jy, 1.0, 0.0, r3 + (size_t)kk(last) * ED, ATOL_DP));
        ps.end();
        LKCHK(fetch(c, 0, 2 * npan + 1, RS));
        auto host = [&](double *dev) { return c->red_host + (dev - c->red); };
        if (h)
            for (int p = 0; p < npan; ++p)
                for (int i = 0; i < kk(p) * ED; ++i)
                    h[(size_t)c0(p) * ED + i] = two_pass ? host(r1(p))[i] + host(r2(p))[i] : host(r1(p))[i];   // gram_schmidt.fypp:49
        n0 = host(r1(0))[kk(0) * ED];
        n2 = host(r3)[kk(last) * ED];
        n1 = two_pass ? host(r2(last))[kk(last) * ED] : n2;
    } else {
        // wider still: column panels of KMAX_FUSED, unfused schedule with a host round trip per panel
        std::vector<double> hacc((size_t)k * ED, 0.0), hp((size_t)k * ED);
        const int npass = two_pass ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            for (int c0 = 0; c0 < k; c0 += KMAX_FUSED) {
                const int kk = (k - c0) < KMAX_FUSED ? (k - c0) : KMAX_FUSED;
                LKCHK((sweepm<1>(Bx, c0, kk, y, nullptr, nullptr, 0, c->red)));
                LKCHK(fetch(c, 0, 1));
                memcpy(hp.data() + (size_t)c0 * ED, c->red_host, (size_t)kk * ED * sizeof(double));
                if (c0 == 0) (pass == 0 ? n0 : n1) = c->red_host[kk * ED];
            }
            for (int c0 = 0; c0 < k; c0 += KMAX_FUSED) {
                const int kk = (k - c0) < KMAX_FUSED ? (k - c0) : KMAX_FUSED;
                HIPCHK(hipMemcpyAsync(c->coef, hp.data() + (size_t)c0 * ED, (size_t)kk * ED * sizeof(double),
                                      hipMemcpyHostToDevice, c->stream));
                LKCHK((sweepm<3>(Bx, c0, kk, y, c->coef, nullptr, 1, c->red + 2 * RED_SECTION)));
                LKCHK(fetch(c, 2, 1));
                (pass == 0 && two_pass ? n1 : n2) = c->red_host[2 * RED_SECTION + kk * ED];
            }
            for (size_t i = 0; i < hacc.size(); ++i) hacc[i] += hp[i];
        }
        if (!two_pass) n1 = n2;
        if (flags & LK_DGS_NORMALIZE) {
            const double nr = std::sqrt(std::fabs(n2));
            if (nr >= ATOL_DP) LKCHK(scal_launch(By, jy, 1.0 / nr, 0.0, nullptr, 0.0));
        }
        if (h) memcpy(h, hacc.data(), hacc.size() * sizeof(double));
    }
    const double s0 = std::sqrt(std::fabs(n0)), s1 = std::sqrt(std::fabs(n1)), s2 = std::sqrt(std::fabs(n2));
    if (norms) { norms[0] = s0; norms[1] = s1; norms[2] = s2; }
    // zero-vector flag of the LAST pass executed (gram_schmidt.fypp:126-127; pass 2 overwrites pass 1)
    if (info) *info = ((two_pass ? s1 : s0) < ATOL_DP) ? 1 : 0;
    if (s2 != s2) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
    return LK_OK;
}

int lk_orthogonalize(lk_basis_t Bx, int k, lk_basis_t By, int jy, double *h, int *info) {
    return dgs_generic(Bx, k, By, jy, h, nullptr, 0, info, false);
}

int lk_dgs(lk_basis_t Bx, int k, lk_basis_t By, int jy, double *h, double *norms, int flags, int *info) {
    return dgs_generic(Bx, k, By, jy, h, norms, flags, info, true);
}

// Y(:, jy0:jy0+qn) -= X(:, :k) * C, C = device coefficients laid out [q][ldc][ED] (what the multi-RHS dot sweep leaves in c->red)
static int gemm_subtract(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int qn, const double *Cdev, int64_t ldc, int c0 = 0) {
    lk_context_t c = Bx->ctx;
    LKCHK(ensure_scratch(c, gemm_packed_doubles(k, qn, Bx->ed())));
    return gemm_launch(Bx, c0, k, By, jy0, qn, Cdev, ldc, -1.0, 1, c->scratch);
}

// ---- block Gram-Schmidt: the panel x panel schedule, split into "enqueue" and "collect" so that one call (lk_dgs_block) or a whole
// block Arnoldi batch (lk_arnoldi_block) synchronises ONCE ------------------------------------------------------------------------
// DGS_basis_against_basis (gram_schmidt.fypp:59-105):  H1 = X^H Y | Y -= X H1 | H2 = X^H Y | Y -= X H2, up to 32 / 4 columns of Y per
// pass over X.  The coefficient sections of every group land in a device slot (compact: per group, per column panel of X, pass 1
// then pass 2, pn * (kk + 1) * ED doubles each, slot kk of a column = ||Y_q||^2); `collect` reads a host copy of it.
static bool dgs_block_fused_ok(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int p) {
    return k >= 1 && k <= KMAX_WIDE && k <= Bx->ncols && p >= 2 && Bx->ctx == By->ctx && Bx->dtype == By->dtype && Bx->n == By->n &&
           !(Bx->data == By->data && jy0 < k) && (k <= KMAX_FUSED || Bx->ctx->xhy_mfma);
}
static bool dgs_block_on_mfma(lk_basis_t Bx, int k, int p) { return Bx->ctx->xhy_mfma && (p >= XHY_MIN_P || k > KMAX_FUSED); }
static int64_t dgs_block_slot_doubles(lk_basis_t Bx, int k, int p) {
    const int ED = Bx->ed();
    if (dgs_block_on_mfma(Bx, k, p)) {
        const int npan = (k + XHY_MAX - 1) / XHY_MAX;
        return (int64_t)2 * p * (k + npan) * ED;
    }
    return (int64_t)2 * p * (k + 1) * ED;
}

// mode 0: enqueue the kernels of every group and the device-to-device copies of its coefficient sections into slot_dev
// mode 1: read a host copy of the slot: h (k x p column-major, may be NULL), *info (last zero column, pass 2: gram_schmidt.fypp:171-173)
static int dgs_block_walk(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int p, double *slot_dev, const double *slot_host, double *h, int *info,
                          int mode) {
    lk_context_t c = Bx->ctx;
    const int ED = Bx->ed();
    int inf = 0;
    int64_t off = 0;
    auto put = [&](const double *src, int64_t cnt) -> int {       // (mode 0) section -> slot
        HIPCHK(hipMemcpyAsync(slot_dev + off, src, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return LK_OK;
    };
    if (dgs_block_on_mfma(Bx, k, p)) {
        // many right-hand sides: coefficients AND updates on the matrix cores, THREE passes over X per group of up to 32 columns of Y
        // (H1 = X^H Y | Y' = Y - X H1 with H2 = X^H Y' in the same pass | Y'' = Y' - X H2); "block_fused" = 0 keeps four passes.  A basis
        // wider than 128 columns runs the same schedule over column PANELS of X (<= 128 columns each, coefficients in a slot of their own):
        //   A: H1_c = X_c^H Y for every panel c | B: Y -= X_c H1_c, the LAST panel's update fused with H2_last = X_last^H Y' and ||Y'||^2 |
        //   C: H2_c = X_c^H Y' for the other panels | D: Y' -= X_c H2_c for every panel     = 4k - |last| columns of X per group
        const int npan = (k + XHY_MAX - 1) / XHY_MAX, last = npan - 1;
        auto pc0 = [&](int cp_) { return cp_ * XHY_MAX; };
        auto pkk = [&](int cp_) { return (k - pc0(cp_)) < XHY_MAX ? (k - pc0(cp_)) : XHY_MAX; };
        const bool fused = c->block_fused == 2 || (c->block_fused && Bx->dtype == LK_F64);   // complex: the 4-pass schedule is faster (2 = force)
        double *o1[XHY_SLOTS / 2] = {nullptr, nullptr, nullptr, nullptr}, *o2[XHY_SLOTS / 2] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < p; j += XHY_GROUP) {
            const int pn = (p - j) < XHY_GROUP ? (p - j) : XHY_GROUP;
            if (mode == 0) {
                for (int cp_ = 0; cp_ < npan; ++cp_)                                                // A
                    LKCHK(dots_mfma(Bx, pc0(cp_), pkk(cp_), By, jy0 + j, pn, 0, cp_, &o1[cp_], cp_ == 0));
                for (int cp_ = 0; cp_ < last; ++cp_)                                                // B
                    LKCHK(gemm_subtract(Bx, pkk(cp_), By, jy0 + j, pn, o1[cp_], (int64_t)(pkk(cp_) + 1), pc0(cp_)));
                if (fused) {
                    LKCHK(upd_dots_mfma(Bx, pc0(last), pkk(last), By, jy0 + j, pn, o1[last], XHY_SLOTS / 2 + last, &o2[last]));
                } else {
                    LKCHK(gemm_subtract(Bx, pkk(last), By, jy0 + j, pn, o1[last], (int64_t)(pkk(last) + 1), pc0(last)));
                    LKCHK(dots_mfma(Bx, pc0(last), pkk(last), By, jy0 + j, pn, 0, XHY_SLOTS / 2 + last, &o2[last], false));
                }
                for (int cp_ = 0; cp_ < last; ++cp_)                                                // C
                    LKCHK(dots_mfma(Bx, pc0(cp_), pkk(cp_), By, jy0 + j, pn, 0, XHY_SLOTS / 2 + cp_, &o2[cp_], false));
                for (int cp_ = 0; cp_ < npan; ++cp_)                                                // D
                    LKCHK(gemm_subtract(Bx, pkk(cp_), By, jy0 + j, pn, o2[cp_], (int64_t)(pkk(cp_) + 1), pc0(cp_)));
            }
            // sections of this group: [panel][pass]
            std::vector<int64_t> o1off(npan), o2off(npan);
            for (int cp_ = 0; cp_ < npan; ++cp_) {
                const int64_t cnt = (int64_t)pn * (pkk(cp_) + 1) * ED;
                o1off[cp_] = off;
                if (mode == 0) LKCHK(put(o1[cp_], cnt));
                off += cnt;
                o2off[cp_] = off;
                if (mode == 0) LKCHK(put(o2[cp_], cnt));
                off += cnt;
            }
            if (mode == 1) {
                for (int q = 0; q < pn; ++q) {
                    // ||Y_q||^2 rides in slot kk of every panel's column q: before pass 1 from panel 0, of the finished Y' from the last panel
                    const int k0 = pkk(0), kl = pkk(last);
                    const double n1 = std::sqrt(std::fabs(slot_host[o1off[0] + ((int64_t)q * (k0 + 1) + k0) * ED]));
                    const double n2 = std::sqrt(std::fabs(slot_host[o2off[last] + ((int64_t)q * (kl + 1) + kl) * ED]));
                    if (n2 < ATOL_DP) inf = j + q + 1;                              // gram_schmidt.fypp:171-173 (pass 2 overwrites)
                    if (n1 != n1 || n2 != n2) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
                    if (h)
                        for (int cp_ = 0; cp_ < npan; ++cp_) {
                            const int kk = pkk(cp_);
                            const double *r1 = slot_host + o1off[cp_] + (int64_t)q * (kk + 1) * ED;
                            const double *r2 = slot_host + o2off[cp_] + (int64_t)q * (kk + 1) * ED;
                            double *hq = h + ((size_t)(j + q) * k + pc0(cp_)) * ED;
                            for (int i = 0; i < kk * ED; ++i) hq[i] = r1[i] + r2[i];                                             // :97
                        }
                }
            }
        }
        if (info) *info = inf;
        return LK_OK;
    }
    for (int j = 0; j < p; j += 4) {
        const int pn = (p - j) < 4 ? (p - j) : 4;
        double *out1 = c->red, *out2 = c->red + (size_t)RED_MULTI * RED_SECTION;
        const bool cpx = Bx->dtype == LK_C128;
        const int64_t cnt = (int64_t)pn * (k + 1) * ED;
        if (mode == 0) {
            if (c->block_fused && (pn <= 2 || (!cpx && k <= 64))) {
                // THREE passes over X for the group: multi-right-hand-side dots, then the fused update + dot sweep (Y' stays in registers),
                // then the two-coefficient update that writes Y''
                LKCHK(dots_p(Bx, 0, k, By, jy0 + j, pn, out1));
                if (cpx) LKCHK((block_sweeps<true, 16, 8, 2>(Bx, k, By, jy0 + j, pn, out1, out2)));   // 8 waves x 16 columns (16 x 8 on 1024 threads spilled)
                else if (pn <= 2) LKCHK((block_sweeps<false, 16, 8, 2>(Bx, k, By, jy0 + j, pn, out1, out2)));
                else LKCHK((block_sweeps<false, 8, 8, 4>(Bx, k, By, jy0 + j, pn, out1, out2)));   // 4 right-hand sides x 8 columns per wave: k <= 64
            } else {
                for (int pass = 0; pass < 2; ++pass) {           // dots / MFMA update / dots / MFMA update: four passes
                    double *out = pass == 0 ? out1 : out2;
                    LKCHK(dots_p(Bx, 0, k, By, jy0 + j, pn, out));
                    LKCHK(gemm_subtract(Bx, k, By, jy0 + j, pn, out, (int64_t)(k + 1)));
                }
            }
            LKCHK(put(out1, cnt));
        }
        const double *r1 = slot_host ? slot_host + off : nullptr;
        off += cnt;
        if (mode == 0) LKCHK(put(out2, cnt));
        const double *r2 = slot_host ? slot_host + off : nullptr;
        off += cnt;
        if (mode == 1) {
            for (int q = 0; q < pn; ++q) {
                const double n1 = std::sqrt(std::fabs(r1[((size_t)q * (k + 1) + k) * ED]));
                const double n2 = std::sqrt(std::fabs(r2[((size_t)q * (k + 1) + k) * ED]));
                if (n2 < ATOL_DP) inf = j + q + 1;                              // gram_schmidt.fypp:171-173 (pass 2 overwrites)
                if (n1 != n1 || n2 != n2) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
                if (h)
                    for (int i = 0; i < k * ED; ++i)
                        h[((size_t)(j + q) * k) * ED + i] = r1[(size_t)q * (k + 1) * ED + i] + r2[(size_t)q * (k + 1) * ED + i];   // :97
            }
        }
    }
    if (info) *info = inf;
    return LK_OK;
}

static int ensure_block_buffers(lk_context_t c, int64_t doubles) {
    if (c->blk_cap >= doubles) return LK_OK;
    if (c->blk_red) HIPCHK(hipFree(c->blk_red));
    if (c->blk_red_host) HIPCHK(hipHostFree(c->blk_red_host));
    c->blk_red = nullptr; c->blk_red_host = nullptr; c->blk_cap = 0;
    HIPCHK(hipMalloc((void **)&c->blk_red, (size_t)doubles * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&c->blk_red_host, (size_t)doubles * sizeof(double), hipHostMallocDefault));
    c->blk_cap = doubles;
    return LK_OK;
}

int lk_dgs_block(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int p, double *h, int *info) {
    if (!Bx || !By) return fail(LK_ERR_INVALID, "lk_dgs_block: null basis");
    DevGuard dev_guard(Bx->ctx);
    if (p < 1 || jy0 < 0 || jy0 + p > By->ncols) return fail(LK_ERR_INVALID, "lk_dgs_block: bad column range");
    const int ED = Bx->ed();
    int inf = 0;
    if (dgs_block_fused_ok(Bx, k, By, jy0, p)) {
        // panel x panel schedule; ONE copy + synchronisation per call (round 6; one per group of columns before)
        lk_context_t c = Bx->ctx;
        LKCHK(lazy_enter(c, true));
        const int64_t need = dgs_block_slot_doubles(Bx, k, p);
        LKCHK(ensure_block_buffers(c, need));
        LKCHK(dgs_block_walk(Bx, k, By, jy0, p, c->blk_red, nullptr, nullptr, nullptr, 0));
        HIPCHK(hipMemcpyAsync(c->blk_red_host, c->blk_red, (size_t)need * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->prof) prof_collect(c);
        return dgs_block_walk(Bx, k, By, jy0, p, nullptr, c->blk_red_host, h, info, 1);
    }
    for (int j = 0; j < p; ++j) {
        int ij = 0;
        LKCHK(dgs_generic(Bx, k, By, jy0 + j, h ? h + (size_t)j * k * ED : nullptr, nullptr, 0, &ij, true));
        if (ij) inf = j + 1;  // `info = i` for the last zero column   gram_schmidt.fypp:171-173
    }
    if (info) *info = inf;
    return LK_OK;
}

// ---- operators ------------------------------------------------------------------------------------
int lk_linop_diag_create(lk_context_t c, int dtype, int64_t n_local, const void *d_host, lk_linop_t *op) {
    if (!c || !d_host || !op) return fail(LK_ERR_INVALID, "lk_linop_diag_create: null argument");
    DevGuard dev_guard(c);
    if (dtype != LK_F64 && dtype != LK_C128) return fail(LK_ERR_INVALID, "bad dtype");
    lk_linop_t o = new lk_linop_s();
    o->ctx = c; o->kind = OP_DIAG; o->dtype = dtype; o->n = n_local;
    const size_t bytes = (size_t)(n_local + 2) * (dtype == LK_C128 ? 2 : 1) * sizeof(double);
    hipError_t e = hipMalloc((void **)&o->dev, bytes);
    if (e != hipSuccess) { delete o; return fail(LK_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    e = hipMemcpy(o->dev, d_host, (size_t)n_local * (dtype == LK_C128 ? 2 : 1) * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(o->dev); delete o; return fail(LK_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e)); }
    *op = o;
    return LK_OK;
}

int lk_linop_diag_linspace_create(lk_context_t c, int64_t n_local, int64_t row0, double d0, double dstep, lk_linop_t *op) {
    if (!c || !op) return fail(LK_ERR_INVALID, "null argument");
    lk_linop_t o = new lk_linop_s();
    o->ctx = c; o->kind = OP_DIAG_LIN; o->dtype = LK_F64; o->n = n_local; o->row0 = row0; o->d0 = d0; o->dstep = dstep;
    *op = o;
    return LK_OK;
}

// layout of a row partition for the all-gather of x (counts and displacements in doubles) + the workspaces of a sharded operator
static int shard_setup(lk_linop_t o, lk_context_t c, int64_t n_global, const int64_t *row_starts, const char *what) {
    const int ED = o->dtype == LK_C128 ? 2 : 1;
    if (!row_starts) return fail(LK_ERR_INVALID, "%s: null row_starts", what);
    if (row_starts[0] != 0 || row_starts[c->nranks] != n_global) return fail(LK_ERR_INVALID, "%s: row_starts must run from 0 to n_global", what);
    for (int r = 0; r < c->nranks; ++r)
        if (row_starts[r + 1] < row_starts[r]) return fail(LK_ERR_INVALID, "%s: row_starts decreases at rank %d", what, r);
    o->ncols_g = n_global;
    o->n = row_starts[c->rank + 1] - row_starts[c->rank];
    o->row0 = row_starts[c->rank];
    o->gcounts.resize(c->nranks);
    o->gdispls.resize(c->nranks);
    for (int r = 0; r < c->nranks; ++r) {
        o->gcounts[r] = (row_starts[r + 1] - row_starts[r]) * ED;
        o->gdispls[r] = row_starts[r] * ED;
    }
    if (c->nranks > 1) HIPCHK(hipMalloc((void **)&o->xfull, (size_t)(n_global > 0 ? n_global : 1) * ED * sizeof(double)));
    return LK_OK;
}

constexpr int64_t GEMV_COLS_PER_CHUNK = 256;

static int dense_finish(lk_linop_t o, lk_context_t c) {
    // split-K of y = A x in chunks of GEMV_COLS_PER_CHUNK columns: the chunking depends on the GLOBAL column count only, so a
    // row's products are added in the same order whatever the row partition (sharded matvec = single-rank matvec bit for bit)
    (void)c;
    int64_t chunks = (o->ncols_g + GEMV_COLS_PER_CHUNK - 1) / GEMV_COLS_PER_CHUNK;
    if (chunks < 1) chunks = 1;
    o->gchunks = (int)chunks;
    const int ED = o->dtype == LK_C128 ? 2 : 1;
    HIPCHK(hipMalloc((void **)&o->gpart, (size_t)chunks * (o->n > 0 ? o->n : 1) * ED * sizeof(double)));
    return LK_OK;
}

int lk_linop_dense_wrap_sharded(lk_context_t c, int dtype, int64_t n_global, const int64_t *row_starts, void *dev_ptr, int64_t lda,
                                lk_linop_t *op) {
    if (!c || !dev_ptr || !op) return fail(LK_ERR_INVALID, "lk_linop_dense_wrap_sharded: null argument");
    DevGuard dev_guard(c);
    if (dtype != LK_F64 && dtype != LK_C128) return fail(LK_ERR_INVALID, "bad dtype");
    if (((uintptr_t)dev_ptr & 15) != 0) return fail(LK_ERR_INVALID, "lk_linop_dense_wrap_sharded: pointer must be 16-byte aligned");
    if (dtype == LK_F64 && (lda & 1)) return fail(LK_ERR_INVALID, "lk_linop_dense_wrap_sharded: lda must be even for LK_F64");
    lk_linop_t o = new lk_linop_s();
    o->ctx = c; o->kind = OP_DENSE; o->dtype = dtype; o->dev = (double *)dev_ptr; o->own_dev = false; o->lda = lda;
    int rc = shard_setup(o, c, n_global, row_starts, "lk_linop_dense_wrap_sharded");
    if (rc == LK_OK && lda < o->n) rc = fail(LK_ERR_INVALID, "lk_linop_dense_wrap_sharded: lda < local rows");
    if (rc == LK_OK) rc = dense_finish(o, c);
    if (rc != LK_OK) { (void)lk_linop_destroy(o); return rc; }
    *op = o;
    return LK_OK;
}

int lk_linop_dense_create_sharded(lk_context_t c, int dtype, int64_t n_global, const int64_t *row_starts, const void *A_rows, int64_t lda,
                                  lk_linop_t *op) {
    if (!c || !A_rows || !op || !row_starts) return fail(LK_ERR_INVALID, "lk_linop_dense_create_sharded: null argument");
    DevGuard dev_guard(c);
    if (dtype != LK_F64 && dtype != LK_C128) return fail(LK_ERR_INVALID, "bad dtype");
    const int64_t n_local = row_starts[c->rank + 1] - row_starts[c->rank];
    if (n_local < 0 || lda < n_local) return fail(LK_ERR_INVALID, "lk_linop_dense_create_sharded: lda < local rows");
    const size_t es = (dtype == LK_C128 ? 2 : 1) * sizeof(double);
    const int64_t ldd = ((n_local + 31) / 32) * 32 + (n_local == 0 ? 32 : 0);          // 256-byte columns on the device
    double *dev = nullptr;
    hipError_t e = hipMalloc((void **)&dev, (size_t)ldd * (n_global > 0 ? n_global : 1) * es);
    if (e != hipSuccess) return fail(LK_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    if (n_local > 0 && n_global > 0) {
        e = hipMemcpy2D(dev, (size_t)ldd * es, A_rows, (size_t)lda * es, (size_t)n_local * es, n_global, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(dev); return fail(LK_ERR_HIP, "hipMemcpy2D failed: %s", hipGetErrorString(e)); }
    }
    const int rc = lk_linop_dense_wrap_sharded(c, dtype, n_global, row_starts, dev, ldd, op);
    if (rc != LK_OK) { (void)hipFree(dev); return rc; }
    (*op)->own_dev = true;
    return LK_OK;
}

int lk_linop_dense_create(lk_context_t c, int dtype, int64_t n, const void *A_host, int64_t lda, lk_linop_t *op) {
    if (!c || !A_host || !op) return fail(LK_ERR_INVALID, "lk_linop_dense_create: null argument");
    if (lda < n) return fail(LK_ERR_INVALID, "lda < n");
    if (c->nranks > 1) return fail(LK_ERR_INVALID, "dense_linop over several ranks: use lk_linop_dense_create_sharded (a row block per rank)");
    const int64_t starts[2] = {0, n};
    return lk_linop_dense_create_sharded(c, dtype, n, starts, A_host, lda, op);
}

// x (this rank's rows) -> the operator's full-length buffer, rank r's block at its global offset
static int gather_x(lk_linop_t op, const double *x, const double **xg) {
    lk_context_t c = op->ctx;
    if (c->nranks == 1) { *xg = x; return LK_OK; }
    if (!c->allgather) return fail(LK_ERR_COMM, "row-sharded dense / CSR operator but no all-gather installed (lk_comm_init_rank / lk_set_allgather)");
    ProfScope ps(c, "comm_allgather", (double)op->ncols_g * (op->dtype == LK_C128 ? 16.0 : 8.0));
    const int rc = c->allgather(c->allgather_user, x, op->xfull, op->gcounts.data(), op->gdispls.data(), c->nranks, (void *)c->stream);
    if (rc != 0) return fail(LK_ERR_COMM, "all-gather callback returned %d", rc);
    *xg = op->xfull;
    return LK_OK;
}

// adjoint products of a row block are full-length partial sums: summed over the ranks, then this rank keeps its own rows
static int reduce_to_slice(lk_linop_t op, double *zfull, double *y) {
    lk_context_t c = op->ctx;
    const int ED = op->dtype == LK_C128 ? 2 : 1;
    LKCHK(allreduce(c, zfull, op->ncols_g * ED));
    const int64_t nd = op->n * ED;
    hipLaunchKernelGGL(k_copy_guarded, dim3(blas1_grid(c, nd / 2 + 1)), dim3(256), 0, c->stream, (const double *)(zfull + op->row0 * ED), y, nd, c->guard());
    HIPCHK(hipGetLastError());
    return LK_OK;
}

int lk_set_allgather(lk_context_t c, lk_allgather_fn fn, void *user) {
    if (!c) return fail(LK_ERR_INVALID, "lk_set_allgather: null context");
    c->allgather = fn;
    c->allgather_user = user;
    return LK_OK;
}

static int halo_exchange(lk_context_t c, const double *send_lo, const double *send_hi, double *recv_lo, double *recv_hi, int64_t count) {
    if (!c->halo) return fail(LK_ERR_COMM, "row-sharded stencil operator but no halo exchange installed (lk_comm_init_rank / lk_set_halo_exchange)");
    ProfScope ps(c, "comm_halo", (double)count * 8.0 * ((send_lo ? 1 : 0) + (send_hi ? 1 : 0)));
    const int rc = c->halo(c->halo_user, send_lo, send_hi, recv_lo, recv_hi, count, (void *)c->stream);
    if (rc != 0) return fail(LK_ERR_COMM, "halo exchange callback returned %d", rc);
    return LK_OK;
}

int lk_set_halo_exchange(lk_context_t c, lk_halo_fn fn, void *user) {
    if (!c) return fail(LK_ERR_INVALID, "lk_set_halo_exchange: null context");
    c->halo = fn;
    c->halo_user = user;
    return LK_OK;
}

int lk_linop_lap5_create_sharded(lk_context_t c, int64_t N, int64_t j0, int64_t nj, lk_linop_t *op) {
    if (!c || !op || N < 1 || j0 < 0 || nj < 1 || j0 + nj > N) return fail(LK_ERR_INVALID, "lk_linop_lap5_create: bad argument");
    DevGuard dev_guard(c);
    lk_linop_t o = new lk_linop_s();
    o->ctx = c; o->kind = OP_LAP5; o->dtype = LK_F64; o->n = nj * N; o->N = N; o->NJ = nj;
    o->has_lo = j0 > 0; o->has_hi = j0 + nj < N;
    // the halo exchange addresses its peers by RANK ORDER (rank - 1 below, rank + 1 above): a partition that does not follow
    // it would skip a send on one side and leave the neighbour waiting
    if (c->nranks > 1 && (o->has_lo != (c->rank > 0) || o->has_hi != (c->rank < c->nranks - 1))) {
        delete o;
        return fail(LK_ERR_INVALID, "lk_linop_lap5_create_sharded: grid lines [%lld, %lld) of %lld on rank %d/%d do not follow rank order "
                    "(rank r must own the r-th consecutive block)", (long long)j0, (long long)(j0 + nj), (long long)N, c->rank, c->nranks);
    }
    if (o->has_lo || o->has_hi) {
        hipError_t e = hipMalloc((void **)&o->halo, (size_t)2 * N * sizeof(double));
        if (e != hipSuccess) { delete o; return fail(LK_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    }
    *op = o;
    return LK_OK;
}

int lk_linop_lap5_create(lk_context_t c, int64_t N, lk_linop_t *op) {
    if (c && c->nranks > 1) return fail(LK_ERR_INVALID, "lap5 over several ranks: use lk_linop_lap5_create_sharded (grid lines per rank)");
    return lk_linop_lap5_create_sharded(c, N, 0, N, op);
}

int lk_linop_gl_create_sharded(lk_context_t c, int64_t n_global, int64_t row0, int64_t n_local, double dx, double tau, int nsub,
                               const double *nu, const double *gamma, double mu_c, double mu2, lk_linop_t *op) {
    if (!c || !op || !nu || !gamma || n_global < 1 || n_local < 1 || row0 < 0 || row0 + n_local > n_global || nsub < 1 || !(dx > 0.0))
        return fail(LK_ERR_INVALID, "lk_linop_gl_create: bad argument");
    DevGuard dev_guard(c);
    lk_linop_t o = new lk_linop_s();
    o->ctx = c; o->kind = OP_GL; o->dtype = LK_C128; o->n = n_local; o->tau = tau; o->nsub = nsub;
    o->row0 = row0; o->n_global = n_global;
    o->has_lo = row0 > 0; o->has_hi = row0 + n_local < n_global;
    if (c->nranks > 1 && (o->has_lo != (c->rank > 0) || o->has_hi != (c->rank < c->nranks - 1))) {
        delete o;
        return fail(LK_ERR_INVALID, "lk_linop_gl_create_sharded: rows [%lld, %lld) of %lld on rank %d/%d do not follow rank order "
                    "(rank r must own the r-th consecutive block)", (long long)row0, (long long)(row0 + n_local), (long long)n_global, c->rank, c->nranks);
    }
    o->gl[0] = dx; o->gl[1] = 0.5 * dx * (double)(n_global + 1);   // L = dx (n+1), x = linspace(-L/2, L/2, n+2)
    o->gl[2] = nu[0]; o->gl[3] = nu[1]; o->gl[4] = gamma[0]; o->gl[5] = gamma[1]; o->gl[6] = mu_c; o->gl[7] = mu2;
    hipError_t e = hipMalloc((void **)&o->wk, ((size_t)3 * n_local * 2 + 8) * sizeof(double));
    if (e != hipSuccess) { delete o; return fail(LK_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    o->halo = o->wk + (size_t)6 * n_local;      // 4 doubles
    o->edges = o->halo + 4;                     // 4 doubles
    (void)hipMemsetAsync(o->halo, 0, 8 * sizeof(double), c->stream);
    *op = o;
    return LK_OK;
}

int lk_linop_gl_create(lk_context_t c, int64_t n, double dx, double tau, int nsub, const double *nu, const double *gamma,
                       double mu_c, double mu2, lk_linop_t *op) {
    if (c && c->nranks > 1) return fail(LK_ERR_INVALID, "Ginzburg-Landau operator over several ranks: use lk_linop_gl_create_sharded");
    return lk_linop_gl_create_sharded(c, n, 0, n, dx, tau, nsub, nu, gamma, mu_c, mu2, op);
}

// A = (rowptr, colind, vals) in CSR, 0-based; `which` = 0 stores A itself, 1 its conjugate transpose (built on the host by a
// counting sort: column indices within a row of A^H come out ascending, so both products sum in index order).
static int csr_upload(lk_linop_t o, int which, int64_t n, const int64_t *rowptr, const int32_t *colind, const double *vals, int ED) {
    const int64_t nnz = rowptr[n];
    auto &m = o->csr[which];
    HIPCHK(hipMalloc((void **)&m.rowptr, (size_t)(n + 1) * sizeof(int64_t)));
    HIPCHK(hipMalloc((void **)&m.colind, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t)));
    HIPCHK(hipMalloc((void **)&m.vals, (size_t)(nnz > 0 ? nnz : 1) * ED * sizeof(double)));
    HIPCHK(hipMemcpy(m.rowptr, rowptr, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (nnz > 0) {
        HIPCHK(hipMemcpy(m.colind, colind, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(m.vals, vals, (size_t)nnz * ED * sizeof(double), hipMemcpyHostToDevice));
    }
    {   // CSR-stream partition: consecutive rows, at most CSR_NNZ entries and 256 rows per block; a longer row stands alone
        std::vector<int64_t> rb;
        rb.push_back(0);
        int64_t r = 0;
        while (r < n) {
            int64_t e = r + 1;                                              // at least one row, however long
            while (e < n && e - r < 256 && rowptr[e + 1] - rowptr[r] <= CSR_NNZ) ++e;
            rb.push_back(e);
            r = e;
        }
        m.nblocks = (int64_t)rb.size() - 1;
        HIPCHK(hipMalloc((void **)&m.rowblocks, rb.size() * sizeof(int64_t)));
        HIPCHK(hipMemcpy(m.rowblocks, rb.data(), rb.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    m.nnz_hint = nnz;
    const double mean = n > 0 ? (double)nnz / (double)n : 0.0;
    int W = 1;                                   // lanes per row: the power of two at or below half the mean row length
    while (W < 64 && 2 * W <= mean / 2.0) W <<= 1;   // (5-point Laplacian, n = 1.7e7: W = 1 / 2 / 4 / 8 / 16 -> see DESIGN.md)
    m.W = W;
    return LK_OK;
}

// creation-time metadata exchange through the data-path hook: every rank contributes `mine` (counts[rank] doubles), all receive all
static int host_allgatherv(lk_context_t c, const std::vector<double> &mine, const std::vector<int64_t> &counts, std::vector<double> &all) {
    std::vector<int64_t> displs(c->nranks, 0);
    int64_t total = 0;
    for (int r = 0; r < c->nranks; ++r) { displs[r] = total; total += counts[r]; }
    all.assign((size_t)total, 0.0);
    if (total == 0) return LK_OK;
    if (!c->allgather) return fail(LK_ERR_COMM, "row-sharded CSR operator but no all-gather installed (lk_comm_init_rank / lk_set_allgather)");
    double *ds = nullptr, *dr = nullptr;
    HIPCHK(hipMalloc((void **)&ds, (size_t)(mine.size() > 0 ? mine.size() : 1) * sizeof(double)));
    hipError_t e = hipMalloc((void **)&dr, (size_t)total * sizeof(double));
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ds); return fail(LK_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    int rc = LK_OK;
    if (!mine.empty() && hipMemcpyAsync(ds, mine.data(), mine.size() * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(LK_ERR_HIP, "metadata upload failed");
    if (rc == LK_OK && c->allgather(c->allgather_user, ds, dr, counts.data(), displs.data(), c->nranks, (void *)c->stream) != 0)
        rc = fail(LK_ERR_COMM, "all-gather callback failed (metadata)");
    if (rc == LK_OK && hipMemcpyAsync(all.data(), dr, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail(LK_ERR_HIP, "metadata download failed");
    if (rc == LK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(LK_ERR_HIP, "stream synchronisation failed");
    (void)hipFree(ds); (void)hipFree(dr);
    return rc;
}

// A COLLECTIVE creation routine must not leave a rank behind: a rank that returned early (a bad column index, a failed
// allocation) would let the others wait for it in the next exchange for ever.  Every rank contributes its local status; when any
// rank failed, EVERY rank returns an error -- its own where it has one, otherwise one naming the first rank that failed.  One
// tiny all-gather through the data-path hook + a stream synchronisation: creation time only.
static int agree_status(lk_context_t c, int rc_local, const char *what) {
    if (c->nranks <= 1) return rc_local;
    std::vector<double> mine(1, (double)rc_local), all;
    const int xrc = host_allgatherv(c, mine, std::vector<int64_t>((size_t)c->nranks, 1), all);
    if (xrc != LK_OK) return rc_local != LK_OK ? rc_local : xrc;
    if (rc_local != LK_OK) return rc_local;                   // lk_last_error already says why
    for (int r = 0; r < c->nranks; ++r)
        if (all[(size_t)r] != 0.0)
            return fail(LK_ERR_COMM, "%s: rank %d failed (status %d); every rank returns without the operator", what, r, (int)all[(size_t)r]);
    return LK_OK;
}

// Decide and set up the compressed exchange of a row-sharded CSR operator (COLLECTIVE: every rank calls it at creation).  On
// return `cols` holds the column indices csr[0] is uploaded with: remapped to [own rows | packed remote entries] when o->cx, the
// global ones otherwise.  The exchange is compressed when what travels is less than half of x (stencils, banded matrices: a few
// boundary entries per neighbour); matrices whose rows reach everywhere keep the plain all-gather of x.
static int csr_compress_setup(lk_linop_t o, lk_context_t c, const int64_t *row_starts, int64_t n, const int64_t *rowptr, const int32_t *colind,
                              std::vector<int32_t> &cols) {
    const int P = c->nranks, me = c->rank;
    const int ED = o->dtype == LK_C128 ? 2 : 1;
    const int64_t nnz = rowptr[n], row0 = row_starts[me], n_global = row_starts[P];
    cols.assign(colind, colind + nnz);
    if (P == 1) return LK_OK;
    auto owner = [&](int64_t j) {                        // rank whose block holds global row j
        int lo = 0, hi = P - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (row_starts[mid] <= j) lo = mid; else hi = mid - 1; }
        while (lo + 1 < P && row_starts[lo + 1] <= j) ++lo;      // empty blocks share a start
        return lo;
    };
    // what I need from whom: the out-of-block column indices, sorted and unique -- owners hold contiguous row ranges, so the
    // sorted list falls apart into one run per owner (4 bytes per remote entry of transient memory, no per-owner vectors)
    std::vector<int32_t> rem;
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t j = colind[p];
        if (j < row0 || j >= row0 + n) rem.push_back((int32_t)j);
    }
    std::sort(rem.begin(), rem.end());
    rem.erase(std::unique(rem.begin(), rem.end()), rem.end());
    std::vector<std::vector<int64_t>> need(P);
    for (int32_t j : rem) need[owner(j)].push_back(j);
    std::vector<int32_t>().swap(rem);
    // everybody learns everybody's request COUNTS (a P x P table); the lists themselves travel only when the compressed exchange
    // can still pay off: what anybody needs of rank r's block is at least the longest single request for it, so
    // 2 * sum_r max_q count[q][r] >= n_global already means "half of x or more travels anyway" -- decided from the table every
    // rank holds identically, before the O(P * requests) exchange of the lists (matrices whose rows reach widely)
    std::vector<double> mycnt(P), allcnt;
    for (int r = 0; r < P; ++r) mycnt[r] = (double)need[r].size();
    LKCHK(host_allgatherv(c, mycnt, std::vector<int64_t>(P, P), allcnt));
    int64_t maxn = 0;
    for (int r = 0; r < P; ++r) maxn = std::max(maxn, row_starts[r + 1] - row_starts[r]);
    {
        int64_t lower = 0;
        for (int r = 0; r < P; ++r) {
            int64_t mx = 0;
            for (int q = 0; q < P; ++q) mx = std::max(mx, (int64_t)allcnt[(size_t)q * P + r]);
            lower += mx;
        }
        if (2 * lower >= n_global) return LK_OK;
    }
    std::vector<int64_t> listlen(P, 0);
    for (int q = 0; q < P; ++q) for (int r = 0; r < P; ++r) listlen[q] += (int64_t)allcnt[(size_t)q * P + r];
    std::vector<double> mylist, alllist;
    for (int r = 0; r < P; ++r) for (int64_t j : need[r]) mylist.push_back((double)j);
    LKCHK(host_allgatherv(c, mylist, listlen, alllist));
    // S_r = what anybody needs of rank r's block (sorted, unique): the packed buffer rank r contributes per matvec
    std::vector<std::vector<int64_t>> S(P);
    {
        size_t pos = 0;
        for (int q = 0; q < P; ++q)
            for (int r = 0; r < P; ++r) {
                const int64_t cnt = (int64_t)allcnt[(size_t)q * P + r];
                for (int64_t i = 0; i < cnt; ++i) S[r].push_back((int64_t)alllist[pos + (size_t)i]);
                pos += (size_t)cnt;
            }
        for (auto &v : S) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
    }
    int64_t total = 0;
    for (int r = 0; r < P; ++r) total += (int64_t)S[r].size();
    // nearly all of x travels anyway, or the remapped indices would not fit: plain all-gather.  Decided from quantities every rank
    // holds identically (a rank that chose differently would issue a different collective)
    if (2 * total >= n_global || maxn + total > 2147483647LL) return LK_OK;
    o->cx = true;
    o->cx_total = total;
    o->cx_counts.resize(P); o->cx_displs.resize(P);
    std::vector<int64_t> off(P, 0);
    { int64_t a = 0; for (int r = 0; r < P; ++r) { off[r] = a; o->cx_counts[r] = (int64_t)S[r].size() * ED; o->cx_displs[r] = a * ED; a += (int64_t)S[r].size(); } }
    for (int64_t p = 0; p < nnz; ++p) {
        const int64_t j = colind[p];
        if (j >= row0 && j < row0 + n) { cols[(size_t)p] = (int32_t)(j - row0); continue; }
        const int r = owner(j);
        const int64_t k = std::lower_bound(S[r].begin(), S[r].end(), j) - S[r].begin();
        cols[(size_t)p] = (int32_t)(n + off[r] + k);
    }
    o->cx_nsend = (int64_t)S[me].size();
    std::vector<int32_t> sidx((size_t)(o->cx_nsend > 0 ? o->cx_nsend : 1), 0);
    for (int64_t i = 0; i < o->cx_nsend; ++i) sidx[(size_t)i] = (int32_t)(S[me][(size_t)i] - row0);
    HIPCHK(hipMalloc((void **)&o->cx_send_idx, sidx.size() * sizeof(int32_t)));
    HIPCHK(hipMemcpy(o->cx_send_idx, sidx.data(), sidx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&o->cx_sendbuf, (size_t)(o->cx_nsend > 0 ? o->cx_nsend : 1) * ED * sizeof(double)));
    HIPCHK(hipMalloc((void **)&o->cx_xrem, (size_t)(total > 0 ? total : 1) * ED * sizeof(double)));
    return LK_OK;
}

// local (rank-private) validation of a row block: nothing here talks to another rank
static int csr_validate_rows(lk_context_t c, int dtype, int64_t n_global, const int64_t *row_starts, const int64_t *rowptr, const int32_t *colind,
                             const void *vals) {
    if (!rowptr) return fail(LK_ERR_INVALID, "lk_linop_csr_create: null rowptr (this rank's row block was rejected by the caller)");
    if (dtype != LK_F64 && dtype != LK_C128) return fail(LK_ERR_INVALID, "lk_linop_csr_create: bad dtype %d", dtype);
    if (n_global < 0 || n_global > 2147483647LL) return fail(LK_ERR_INVALID, "lk_linop_csr_create: bad size %lld", (long long)n_global);
    const int64_t n = row_starts[c->rank + 1] - row_starts[c->rank];      // rows held here; column indices are GLOBAL
    if (n < 0) return fail(LK_ERR_INVALID, "lk_linop_csr_create: row_starts decreases");
    if (rowptr[0] != 0) return fail(LK_ERR_INVALID, "lk_linop_csr_create: rowptr must be 0-based");
    for (int64_t i = 0; i < n; ++i)
        if (rowptr[i + 1] < rowptr[i]) return fail(LK_ERR_INVALID, "lk_linop_csr_create: rowptr decreases at row %lld", (long long)i);
    const int64_t nnz = rowptr[n];
    if (nnz > 0 && (!colind || !vals)) return fail(LK_ERR_INVALID, "lk_linop_csr_create: null colind / vals");
    for (int64_t p = 0; p < nnz; ++p)
        if (colind[p] < 0 || colind[p] >= n_global) return fail(LK_ERR_INVALID, "lk_linop_csr_create: column index %d out of range at entry %lld", colind[p], (long long)p);
    return LK_OK;
}

int lk_linop_csr_create_sharded(lk_context_t c, int dtype, int64_t n_global, const int64_t *row_starts, const int64_t *rowptr,
                                const int32_t *colind, const void *vals, lk_linop_t *op) {
    // a null context / result / partition is a programming error of the call site, identical on every rank: plain early return.
    // A null rowptr is NOT: it is how a host-side wrapper that found this rank's input unusable (wrong number of rows, wrong
    // value type) still joins the agreement below, so that every rank fails together (lightkrylov_amd/linops.py).
    if (!c || !op || !row_starts) return fail(LK_ERR_INVALID, "lk_linop_csr_create: null argument");
    DevGuard dev_guard(c);
    // COLLECTIVE from here on.  Everything that can fail on ONE rank only (this rank's rows, this rank's allocations) is agreed
    // on before the next exchange is entered: first the validation of the row block ...
    LKCHK(agree_status(c, csr_validate_rows(c, dtype, n_global, row_starts, rowptr, colind, vals), "lk_linop_csr_create_sharded (validation)"));
    const int64_t n = row_starts[c->rank + 1] - row_starts[c->rank];
    const int64_t nnz = rowptr[n];
    const int ED = dtype == LK_C128 ? 2 : 1;
    const double *v = (const double *)vals;
    // conjugate transpose of the row block (n_global rows, LOCAL column indices) by counting sort over the column indices.  Host
    // allocations of THIS rank (n_global + nnz entries): a bad_alloc must neither cross the C ABI nor leave the peers in the next
    // exchange -- it becomes this rank's status in the "buffers" agreement below
    std::vector<int64_t> tp;
    std::vector<int32_t> tc;
    std::vector<double> tv;
    lk_linop_t o = nullptr;
    int rc_host = LK_OK;
    try {
        tp.assign((size_t)n_global + 1, 0);
        for (int64_t p = 0; p < nnz; ++p) tp[(size_t)colind[p] + 1] += 1;
        for (int64_t j = 0; j < n_global; ++j) tp[(size_t)j + 1] += tp[(size_t)j];
        tc.resize((size_t)(nnz > 0 ? nnz : 1));
        tv.resize((size_t)(nnz > 0 ? nnz : 1) * ED);
        std::vector<int64_t> next(tp.begin(), tp.end() - 1);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                const int64_t q = next[(size_t)colind[p]]++;
                tc[(size_t)q] = (int32_t)i;
                tv[(size_t)q * ED] = v[p * ED];
                if (ED == 2) tv[(size_t)q * 2 + 1] = -v[p * 2 + 1];
            }
        o = new lk_linop_s();
        o->ctx = c; o->kind = OP_CSR; o->dtype = dtype;
    } catch (const std::exception &e) {
        rc_host = fail(LK_ERR_NOMEM, "lk_linop_csr_create_sharded: host allocation failed while transposing this rank's rows (%s)", e.what());
    }
    // ... then this rank's share of the set-up that precedes the metadata exchange (the transpose above, the gathered-x buffer) ...
    int rc = agree_status(c, rc_host != LK_OK ? rc_host : shard_setup(o, c, n_global, row_starts, "lk_linop_csr_create"), "lk_linop_csr_create_sharded (buffers)");
    if (rc == LK_OK) {
        // ... the metadata exchange itself (its decisions come from quantities every rank holds identically; its staging buffers
        // are a few P-length tables and the request lists -- a rank that cannot allocate THOSE is fatal for the job, see header) ...
        try {
            std::vector<int32_t> cols0;
            rc = csr_compress_setup(o, c, row_starts, n, rowptr, colind, cols0);
            if (rc == LK_OK) rc = csr_upload(o, 0, n, rowptr, cols0.data(), v, ED);
            if (rc == LK_OK) rc = csr_upload(o, 1, n_global, tp.data(), tc.data(), tv.data(), ED);
        } catch (const std::exception &e) {
            rc = fail(LK_ERR_NOMEM, "lk_linop_csr_create_sharded: host allocation failed (%s)", e.what());
        }
        // ... and finally the uploads: an operator exists on every rank or on none
        rc = agree_status(c, rc, "lk_linop_csr_create_sharded (upload)");
    }
    if (rc != LK_OK) { if (o) (void)lk_linop_destroy(o); return rc; }
    *op = o;
    return LK_OK;
}

int lk_linop_csr_create(lk_context_t c, int dtype, int64_t n, const int64_t *rowptr, const int32_t *colind, const void *vals,
                        lk_linop_t *op) {
    if (!c || !rowptr || !op) return fail(LK_ERR_INVALID, "lk_linop_csr_create: null argument");
    if (c->nranks > 1) return fail(LK_ERR_INVALID, "lk_linop_csr_create over several ranks: use lk_linop_csr_create_sharded (a row block per rank)");
    const int64_t starts[2] = {0, n};
    return lk_linop_csr_create_sharded(c, dtype, n, starts, rowptr, colind, vals, op);
}

int lk_linop_destroy(lk_linop_t op) {
    if (!op) return LK_OK;
    for (auto &m : op->csr) {
        if (m.rowptr) (void)hipFree(m.rowptr);
        if (m.colind) (void)hipFree(m.colind);
        if (m.vals) (void)hipFree(m.vals);
        if (m.rowblocks) (void)hipFree(m.rowblocks);
    }
    if (op->kind == OP_LAP5 && op->halo) (void)hipFree(op->halo);
    if (op->wk) (void)hipFree(op->wk);
    if (op->xfull) (void)hipFree(op->xfull);
    if (op->cx_send_idx) (void)hipFree(op->cx_send_idx);
    if (op->cx_sendbuf) (void)hipFree(op->cx_sendbuf);
    if (op->cx_xrem) (void)hipFree(op->cx_xrem);
    if (op->gpart) (void)hipFree(op->gpart);
    if (op->dev && op->own_dev) (void)hipFree(op->dev);  // synchronises; the context may already be finalized
    delete op;
    return LK_OK;
}

int lk_linop_apply(lk_linop_t op, int trans, lk_basis_t Bx, int jx, lk_basis_t By, int jy) {
    if (!op) return fail(LK_ERR_INVALID, "lk_linop_apply: null operator");
    DevGuard dev_guard(op->ctx);
    LKCHK(check_vec(Bx, jx, "lk_linop_apply(vec_in)"));
    LKCHK(check_vec(By, jy, "lk_linop_apply(vec_out)"));
    LKCHK(check_pair(Bx, By, "lk_linop_apply"));
    if (Bx->dtype != op->dtype || Bx->n != op->n) return fail(LK_ERR_INVALID, "lk_linop_apply: operator/vector mismatch");
    lk_context_t c = op->ctx;
    const double *x = Bx->col(jx);
    double *y = By->col(jy);
    if (x == y) return fail(LK_ERR_INVALID, "lk_linop_apply: vec_in and vec_out alias");
    {
        const VecRef w{By, jy}, r{Bx, jx};
        LKCHK(lazy_enter_vec(c, &w, true, &r, nullptr));       // vec_out is intent(out): AbstractLinops.fypp:74-87
    }
    By->touch(jy);
    const bool cp = op->dtype == LK_C128;
    const int64_t n = op->n;
    const int64_t nv = n * Bx->ed() / 2 + 1;
    // algorithmic bytes of the operator (diagonal: d, x read, y written; generated diagonal: x read, y written); 0 = not priced
    const double mv_bytes = op->kind == OP_DIAG ? 3.0 * n * Bx->ed() * 8.0 : (op->kind == OP_DIAG_LIN ? 2.0 * n * 8.0 : 0.0);
    // the diagonal operators (one kernel) carry their two events on the dispatch itself, like the sweeps: no marker packets
    const bool one_kernel = op->kind == OP_DIAG || op->kind == OP_DIAG_LIN;
    ProfScope ps(c, "matvec", mv_bytes, one_kernel && c->prof_ext);
    switch (op->kind) {
    case OP_DIAG:
        if (ps.on && ps.ext) {
            if (cp) hipExtLaunchKernelGGL(k_diag<true>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, ps.rec.e0, ps.rec.e1, 0, (const double *)op->dev, x, y, n, (int)(trans == LK_OP_H), c->guard());
            else hipExtLaunchKernelGGL(k_diag<false>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, ps.rec.e0, ps.rec.e1, 0, (const double *)op->dev, x, y, n, 0, c->guard());
            break;
        }
        if (cp) hipLaunchKernelGGL(k_diag<true>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, op->dev, x, y, n, trans == LK_OP_H, c->guard());
        else hipLaunchKernelGGL(k_diag<false>, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, op->dev, x, y, n, 0, c->guard());
        break;
    case OP_DIAG_LIN:
        if (ps.on && ps.ext)
            hipExtLaunchKernelGGL(k_diag_linspace, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, ps.rec.e0, ps.rec.e1, 0, op->d0, op->dstep, op->row0, x, y, n, c->guard());
        else
            hipLaunchKernelGGL(k_diag_linspace, dim3(blas1_grid(c, nv)), dim3(256), 0, c->stream, op->d0, op->dstep, op->row0, x, y, n, c->guard());
        break;
    case OP_DENSE: {
        const int ED = Bx->ed();
        if (trans == LK_OP_N) {
            const double *xg = nullptr;
            LKCHK(gather_x(op, x, &xg));                                                  // sharded: all-gather of x
            const int64_t cpc = GEMV_COLS_PER_CHUNK;
            const dim3 grid((unsigned)((n + (cp ? 255 : 511)) / (cp ? 256 : 512)), (unsigned)op->gchunks);
            if (n > 0) {
                if (cp) hipLaunchKernelGGL(k_gemv_n<true>, grid, dim3(256), 0, c->stream, op->dev, op->lda, n, op->ncols_g, xg, op->gpart, cpc, c->guard());
                else hipLaunchKernelGGL(k_gemv_n<false>, grid, dim3(256), 0, c->stream, op->dev, op->lda, n, op->ncols_g, xg, op->gpart, cpc, c->guard());
                hipLaunchKernelGGL(k_gemv_n_finish, dim3(blas1_grid(c, n * ED)), dim3(256), 0, c->stream, (const double *)op->gpart, n * ED, op->gchunks, y, c->guard());
            }
        } else {
            // y = A^H x.  Sharded: z = A_rows^H x_rows has n_global entries on every rank; their sum over the ranks is the
            // product, of which this rank keeps its rows.
            double *z = c->nranks > 1 ? op->xfull : y;
            if (op->ncols_g > 0) {
                if (cp) hipLaunchKernelGGL(k_gemv_h<true>, dim3((unsigned)((op->ncols_g + 3) / 4)), dim3(256), 0, c->stream, op->dev, op->lda, n, op->ncols_g, x, z, c->guard());
                else hipLaunchKernelGGL(k_gemv_h<false>, dim3((unsigned)((op->ncols_g + 3) / 4)), dim3(256), 0, c->stream, op->dev, op->lda, n, op->ncols_g, x, z, c->guard());
            }
            if (c->nranks > 1) LKCHK(reduce_to_slice(op, z, y));
        }
        break;
    }
    case OP_CSR: {
        auto m = op->csr[trans == LK_OP_N ? 0 : 1];
        // row-sharded: 'N' multiplies this rank's rows with the all-gathered x; 'H' applies the conjugate transpose of the row
        // block (n_global rows) to this rank's x, sums the full-length results over the ranks and keeps this rank's rows
        const bool shard = c->nranks > 1;
        const int64_t nr = trans == LK_OP_N ? n : op->ncols_g;          // rows of the product computed here
        const double *xin = x;
        double *yout = y;
        const double *xrem = nullptr;
        int64_t nloc = INT64_MAX;                                        // column indices below nloc address `xin`, the others `xrem`
        if (shard && trans == LK_OP_N && op->cx) {
            // compressed exchange: pack what other ranks' rows reference, all-gather the packed pieces, multiply [own rows | pieces]
            if (!c->allgather) return fail(LK_ERR_COMM, "row-sharded CSR operator but no all-gather installed (lk_comm_init_rank / lk_set_allgather)");
            if (op->cx_nsend > 0) {
                if (cp) hipLaunchKernelGGL(k_pack<true>, dim3(blas1_grid(c, op->cx_nsend)), dim3(256), 0, c->stream, x, op->cx_send_idx, op->cx_nsend, op->cx_sendbuf, c->guard());
                else hipLaunchKernelGGL(k_pack<false>, dim3(blas1_grid(c, op->cx_nsend)), dim3(256), 0, c->stream, x, op->cx_send_idx, op->cx_nsend, op->cx_sendbuf, c->guard());
                HIPCHK(hipGetLastError());
            }
            {
                ProfScope ps(c, "comm_allgather", (double)op->cx_total * (cp ? 16.0 : 8.0));
                if (c->allgather(c->allgather_user, op->cx_sendbuf, op->cx_xrem, op->cx_counts.data(), op->cx_displs.data(), c->nranks, (void *)c->stream) != 0)
                    return fail(LK_ERR_COMM, "all-gather callback failed");
            }
            xrem = op->cx_xrem;
            nloc = n;
        } else if (shard && trans == LK_OP_N) {
            LKCHK(gather_x(op, x, &xin));
        }
        if (shard && trans != LK_OP_N) yout = op->xfull;
        bool launched = false;
        if (c->csr_stream && nr > 0 && (double)m.nnz_hint / (double)nr <= 32.0 && !c->csr_lanes) {
            // short rows: stream the entries through LDS (k_csr_stream)
            int64_t g = m.nblocks;
            const int64_t cap = (int64_t)c->num_cu * 16;
            if (g > cap) g = cap;
            if (g < 1) g = 1;
            if (cp) hipLaunchKernelGGL(k_csr_stream<true>, dim3((unsigned)g), dim3(256), 0, c->stream, m.rowblocks, m.rowptr, m.colind, m.vals, xin, yout, m.nblocks, c->guard(), xrem, nloc);
            else hipLaunchKernelGGL(k_csr_stream<false>, dim3((unsigned)g), dim3(256), 0, c->stream, m.rowblocks, m.rowptr, m.colind, m.vals, xin, yout, m.nblocks, c->guard(), xrem, nloc);
            launched = true;
        }
        if (!launched) {
            if (c->csr_lanes) m.W = c->csr_lanes;
            const int64_t rows_per_block = 256 / m.W;
            int64_t g = (nr + rows_per_block - 1) / rows_per_block;
            const int64_t cap = (int64_t)c->num_cu * 16;
            if (g > cap) g = cap;
            if (g < 1) g = 1;
#define LK_CSR_LAUNCH(WW)                                                                                                  \
    case WW:                                                                                                               \
        if (cp) hipLaunchKernelGGL((k_csr<true, WW>), dim3((unsigned)g), dim3(256), 0, c->stream, m.rowptr, m.colind, m.vals, xin, yout, nr, c->guard(), xrem, nloc); \
        else hipLaunchKernelGGL((k_csr<false, WW>), dim3((unsigned)g), dim3(256), 0, c->stream, m.rowptr, m.colind, m.vals, xin, yout, nr, c->guard(), xrem, nloc);   \
        break;
            switch (m.W) {
                LK_CSR_LAUNCH(1) LK_CSR_LAUNCH(2) LK_CSR_LAUNCH(4) LK_CSR_LAUNCH(8) LK_CSR_LAUNCH(16) LK_CSR_LAUNCH(32) LK_CSR_LAUNCH(64)
            default: return fail(LK_ERR_INVALID, "internal: CSR lanes per row %d", m.W);
            }
#undef LK_CSR_LAUNCH
        }
        HIPCHK(hipGetLastError());
        if (shard && trans != LK_OP_N) LKCHK(reduce_to_slice(op, yout, y));
        break;
    }
    case OP_GL: {
        // nsub classical RK4 steps of dt = tau/nsub; 4 stage launches per step.
        const double dt = op->tau / op->nsub;
        const double *g = op->gl;
        double *ka = op->wk, *kb = op->wk + 2 * n, *us = op->wk + 4 * n;
        const unsigned grid = (unsigned)((n + 255) / 256);
        const int adj = trans == LK_OP_H;
        const double *uin = x;
        for (int sstep = 0; sstep < op->nsub; ++sstep) {
            // outputs alternate between the work vector and y so that the LAST sub-step lands in y
            double *uout = ((op->nsub - 1 - sstep) % 2 == 0) ? y : us;
            const bool sharded = op->has_lo || op->has_hi;
            const double *halo = sharded ? op->halo : nullptr;
#define GL_STAGE(KPREV, A, KOUT, B, FIRST)                                                                                    \
            do {                                                                                                                   \
                if (sharded) {   /* the stage needs v = u + a*kprev one point beyond either end of this rank's block */          \
                    hipLaunchKernelGGL(k_gl_edges, dim3(1), dim3(64), 0, c->stream, uin, KPREV, A, n, op->edges, c->guard());      \
                    LKCHK(halo_exchange(c, op->has_lo ? op->edges : nullptr, op->has_hi ? op->edges + 2 : nullptr,                \
                                        op->has_lo ? op->halo : nullptr, op->has_hi ? op->halo + 2 : nullptr, 2));               \
                }                                                                                                                  \
                hipLaunchKernelGGL(k_gl_stage, dim3(grid), dim3(256), 0, c->stream, uin, KPREV, A, KOUT, uout, B, FIRST, n, g[0], g[1], \
                                   g[2], g[3], g[4], g[5], g[6], g[7], adj, op->row0, op->n_global, halo, c->guard());            \
            } while (0)
            GL_STAGE((const double *)nullptr, 0.0, ka, dt / 6.0, 1);
            GL_STAGE((const double *)ka, 0.5 * dt, kb, dt / 3.0, 0);
            GL_STAGE((const double *)kb, 0.5 * dt, ka, dt / 3.0, 0);
            GL_STAGE((const double *)ka, dt, kb, dt / 6.0, 0);
#undef GL_STAGE
            uin = uout;
        }
        break;
    }
    case OP_LAP5: {
        const int64_t N = op->N, NJ = op->NJ;
        const double s = (double)(N + 1) * (double)(N + 1);
        const double *lo = nullptr, *hi = nullptr;
        if (op->has_lo || op->has_hi) {
            // first / last local grid line to the neighbouring ranks, theirs into the halo buffer (N doubles each)
            lo = op->has_lo ? op->halo : nullptr;
            hi = op->has_hi ? op->halo + N : nullptr;
            LKCHK(halo_exchange(c, op->has_lo ? x : nullptr, op->has_hi ? x + (NJ - 1) * N : nullptr, op->has_lo ? op->halo : nullptr,
                                op->has_hi ? op->halo + N : nullptr, N));
        }
        dim3 grid((unsigned)((N / 2 + 1 + 255) / 256), (unsigned)NJ);
        if ((N & 1) == 0) {                               // persistent blocks over (line, segment) tiles
            int64_t tiles = NJ * ((N / 2 + 255) / 256), g = (int64_t)c->num_cu * c->lap5_grid_mult;
            grid = dim3((unsigned)(tiles < g ? (tiles > 0 ? tiles : 1) : g), 1);
        }
        hipLaunchKernelGGL(k_lap5, grid, dim3(256), 0, c->stream, x, y, N, NJ, lo, hi, s, c->guard());
        break;
    }
    }
    HIPCHK(hipGetLastError());
    return LK_OK;
}

// ---- Arnoldi -----------------------------------------------------------------------------------------
// One step on the host-synchronous schedule (one round trip per step): used for k > KMAX_FUSED and with
// async_arnoldi = 0.  Returns through *stop: 0 continue, 1 loop must exit (info set).
static int arnoldi_step_sync(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int k, double tol, int trans, std::vector<double> &h,
                             int *info, int *stop) {
    const int ED = X->ed();
    *stop = 0;
    LKCHK(lk_linop_apply(A, trans ? LK_OP_H : LK_OP_N, X, k - 1, X, k));          // arnoldi.fypp:39-47
    double norms[3];
    int dinfo = 0;
    LKCHK(lk_dgs(X, k, X, k, h.data(), norms, LK_DGS_NORMALIZE, &dinfo));          // arnoldi.fypp:50-55
    double *Hk = H + (size_t)(k - 1) * ldh * ED;
    memcpy(Hk, h.data(), (size_t)k * ED * sizeof(double));
    const double beta = norms[2];
    Hk[(size_t)k * ED] = 0.0;
    if (ED == 2) Hk[(size_t)k * ED + 1] = 0.0;
    if (beta < ATOL_DP) {
        // qr_no_pivoting, colinear column: R(1,1) = 0, rand, re-normalise   qr.fypp:146-162
        LKCHK(lk_vec_rand(X, k, 0x5EEDull + (uint64_t)k, X->ctx->row0, 1));
    } else {
        Hk[(size_t)k * ED] = beta;
    }
    if (std::fabs(Hk[(size_t)k * ED]) < tol) {                                     // arnoldi.fypp:58-71
        *info = k;
        *stop = 1;
    }
    return LK_OK;
}

// per-step result slots of an asynchronous batch: nsteps x RED_SECTIONS sections of `stride` doubles (red_stride(last step))
static int ensure_step_buffers(lk_context_t c, int nsteps, int stride) {
    const int64_t need = (int64_t)nsteps * RED_SECTIONS * stride;
    if (c->step_red_cap >= need) return LK_OK;
    if (c->step_red) HIPCHK(hipFree(c->step_red));
    if (c->step_red_host) HIPCHK(hipHostFree(c->step_red_host));
    c->step_red = nullptr; c->step_red_host = nullptr; c->step_red_cap = 0;
    const size_t bytes = (size_t)need * sizeof(double);
    HIPCHK(hipMalloc((void **)&c->step_red, bytes));
    HIPCHK(hipHostMalloc((void **)&c->step_red_host, bytes, hipHostMallocDefault));
    c->step_red_cap = need;
    return LK_OK;
}

// Steps [k0, k1] (all <= KMAX_WIDE) enqueued back to back with NO host round trip: operator, three fused sweeps and
// the normalise of step s write their reduction results into step slot s - k0; the normalise kernel raises the device
// stop flag on breakdown (||y''|| < max(tol, atol_dp)) or NaN, which turns every kernel of a later step into a
// no-op.  One D2H copy + one synchronisation per batch.  *done = last step whose results are valid.
// `seg_last` (nseg ascending step numbers within [k0, k1), or nullptr): SEGMENTED delivery -- behind the last step of every segment the
// batch copies that segment's result slots and the stop flag to the host and records an event; the host keeps the device SEG_LOOKAHEAD
// steps ahead of the segment it is waiting for, waits for the events in order and calls deliver(first, last) for the steps of each segment
// while the device runs the later ones (lk_arnoldi_segments: the per-step host work of a caller -- eigs' Ritz tests -- overlaps the rest of
// the cycle with NO idle gap on the device between segments, which one blocking lk_arnoldi call per segment costs).  deliver returns
// LK_OK, an error, or LK_STOP_REQUESTED (the caller's progress function asked to stop): nothing more is enqueued then -- at most
// SEG_LOOKAHEAD steps beyond the delivered ones have run -- and *cancelled is set.  A device-side stop inside a segment ends the
// deliveries there; the caller processes the stopped step after the final synchronisation, exactly as in the unsegmented case.
constexpr int SEG_LOOKAHEAD = 24;
constexpr int LK_STOP_REQUESTED = 1;
static int arnoldi_batch_async(lk_linop_t A, lk_basis_t X, int k0, int k1, double tol, int trans, int *done, const int *seg_last, int nseg,
                               const std::function<int(int, int)> &deliver, int *delivered_to, bool *cancelled) {
    lk_context_t c = X->ctx;
    const int nsteps = k1 - k0 + 1;
    const int rs = red_stride(k1);                  // one section size for every step of the batch
    LKCHK(ensure_step_buffers(c, nsteps, rs));
    if (nseg > c->seg_cap) {
        if (c->seg_stop_host) HIPCHK(hipHostFree(c->seg_stop_host));
        c->seg_stop_host = nullptr;
        HIPCHK(hipHostMalloc((void **)&c->seg_stop_host, (size_t)nseg * sizeof(int), hipHostMallocDefault));
        while ((int)c->seg_events.size() < nseg) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->seg_events.push_back(e);
        }
        c->seg_cap = nseg;
    }
    HIPCHK(hipMemsetAsync(c->stop_dev, 0, sizeof(int), c->stream));
    const double tol_break = tol > ATOL_DP ? tol : ATOL_DP;
    const int ED = X->ed();
    const size_t slot_doubles = (size_t)RED_SECTIONS * rs;
    int enq = k0;                                    // next step to enqueue
    int si = 0, seg_first = k0;                      // next segment to close, first step of it
    auto enqueue_until = [&](int kto) -> int {
        c->guard_on = true;
        c->prof_sweeps_only = true;
        int rc = LK_OK;
        for (; enq <= kto && rc == LK_OK; ++enq) {
            const int k = enq;
            c->guard_step = k;
            double *slot = c->step_red + (size_t)(k - k0) * slot_doubles;
            rc = lk_linop_apply(A, trans ? LK_OP_H : LK_OP_N, X, k - 1, X, k);
            if (rc != LK_OK) break;
            c->span_first = nullptr;
            const bool single = resident_applies(X, k);
            // cache-resident panel: both passes, the normalise and the breakdown test in ONE launch (lk_resident.hip.h)
            if (single) rc = dgs_resident_launch(X, k, X->col(k), slot, rs, true, ATOL_DP, tol_break, c->stop_dev);
            else {
                resident_note_fallback(X, k);
                rc = dgs_device(X, k, X->col(k), true, slot, rs);
            }
            if (rc != LK_OK) break;
            if (c->prof && c->span_first) {          // "dgs" = first sweep's start .. the end of the last reduction, no events of its own
                ProfRec span;
                span.e0 = c->span_first; span.e1 = c->span_last; span.tag = "dgs"; span.borrowed = true;
                span.bytes = (double)X->n * ED * 8.0 * (3.0 * k + 5.0);
                c->prof_pending.push_back(span);
            }
            if (!single) rc = scal_launch(X, k, 1.0, 0.0, slot + 2 * rs + (size_t)k * ED, ATOL_DP, c->stop_dev, tol_break);
            if (rc == LK_OK && si < nseg && k == seg_last[si]) {
                // close a segment: its slots and the stop flag as it stands now travel to the host behind this step
                const size_t off = (size_t)(seg_first - k0) * slot_doubles;
                const hipError_t e1 = hipMemcpyAsync(c->step_red_host + off, c->step_red + off, (size_t)(k - seg_first + 1) * slot_doubles * sizeof(double),
                                                     hipMemcpyDeviceToHost, c->stream);
                const hipError_t e2 = hipMemcpyAsync(c->seg_stop_host + si, c->stop_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream);
                const hipError_t e3 = hipEventRecord(c->seg_events[si], c->stream);
                if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc = fail(LK_ERR_HIP, "lk_arnoldi_segments: copy behind a segment failed");
                seg_first = k + 1;
                ++si;
            }
        }
        c->guard_on = false;
        c->guard_step = 0;
        c->prof_sweeps_only = false;
        return rc;
    };
    *delivered_to = k0 - 1;
    *cancelled = false;
    bool device_stop = false;
    int first = k0;
    for (int i = 0; i < nseg && !device_stop && !*cancelled; ++i) {
        const int ahead = seg_last[i] + SEG_LOOKAHEAD < k1 ? seg_last[i] + SEG_LOOKAHEAD : k1;
        LKCHK(enqueue_until(ahead));
        HIPCHK(hipEventSynchronize(c->seg_events[i]));
        if (c->seg_stop_host[i] != 0) { device_stop = true; break; }   // handled by the caller after the final synchronisation
        const int drc = deliver(first, seg_last[i]);
        if (drc == LK_STOP_REQUESTED) *cancelled = true;
        else LKCHK(drc);
        *delivered_to = seg_last[i];
        first = seg_last[i] + 1;
    }
    if (!*cancelled && !device_stop) LKCHK(enqueue_until(k1));   // (after a device-side stop every later step would be a no-op: not enqueued)
    const int klast = enq - 1;                       // last step enqueued
    if (klast >= seg_first) {                        // the rest (everything, without segments)
        const size_t off = (size_t)(seg_first - k0) * slot_doubles;
        HIPCHK(hipMemcpyAsync(c->step_red_host + off, c->step_red + off, (size_t)(klast - seg_first + 1) * slot_doubles * sizeof(double),
                              hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipMemcpyAsync(c->stop_host, c->stop_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    const int stop_step = *c->stop_host;
    *done = stop_step ? stop_step : klast;
    return LK_OK;
}

// Lanczos steps [k0, k1] (all <= KMAX_WIDE) enqueued back to back, as arnoldi_batch_async: operator; the two local
// orthogonalisations of update_tridiag_matrix (lanczos.fypp:57-60) -- each T(i, k) = X(i)%dot(X(k+1)) with its
// X(k+1)%axpby(-T(i, k), X(i), 1) is a one-column Gram-Schmidt pass: dot sweep, coefficient left on the device, update sweep --;
// the full re-orthogonalisation (three fused sweeps); the normalise kernel with the device-side stop flag.
static int lanczos_batch_async(lk_linop_t A, lk_basis_t X, int k0, int k1, double tol, int *done) {
    lk_context_t c = X->ctx;
    const int nsteps = k1 - k0 + 1;
    const int rs = red_stride(k1);
    LKCHK(ensure_step_buffers(c, nsteps, rs));
    if (c->lz_cap < nsteps) {
        if (c->lz_red) HIPCHK(hipFree(c->lz_red));
        if (c->lz_red_host) HIPCHK(hipHostFree(c->lz_red_host));
        c->lz_red = nullptr; c->lz_red_host = nullptr; c->lz_cap = 0;
        const size_t bytes = (size_t)nsteps * 4 * RED_SECTION * sizeof(double);
        HIPCHK(hipMalloc((void **)&c->lz_red, bytes));
        HIPCHK(hipHostMalloc((void **)&c->lz_red_host, bytes, hipHostMallocDefault));
        c->lz_cap = nsteps;
    }
    HIPCHK(hipMemsetAsync(c->stop_dev, 0, sizeof(int), c->stream));
    const double tol_break = tol > ATOL_DP ? tol : ATOL_DP;     // below it: stop flag AND no scaling (lanczos.fypp:32-36)
    c->guard_on = true;
    c->prof_sweeps_only = true;
    int rc = LK_OK;
    for (int k = k0; k <= k1 && rc == LK_OK; ++k) {
        c->guard_step = k;
        double *slot = c->step_red + (size_t)(k - k0) * RED_SECTIONS * rs;
        rc = lk_linop_apply(A, LK_OP_N, X, k - 1, X, k);
        const int i0 = k > 1 ? k - 1 : 1;
        for (int i = i0; i <= k && rc == LK_OK; ++i) {                       // lanczos.fypp:57-60
            double *a = c->lz_red + ((size_t)(k - k0) * 4 + 2 * (i - i0)) * RED_SECTION;
            rc = sweepm<1>(X, i - 1, 1, X->col(k), nullptr, nullptr, 1, a);
            if (rc == LK_OK) rc = sweepm<3>(X, i - 1, 1, X->col(k), a, nullptr, 1, a + RED_SECTION);
        }
        if (rc != LK_OK) break;
        rc = dgs_step_async(X, k, X, k, slot, rs, tol_break, tol_break);   // :62 (no beta), :29-39
    }
    c->guard_on = false;
    c->guard_step = 0;
    c->prof_sweeps_only = false;
    LKCHK(rc);
    HIPCHK(hipMemcpyAsync(c->step_red_host, c->step_red, (size_t)nsteps * RED_SECTIONS * rs * sizeof(double),
                          hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->lz_red_host, c->lz_red, (size_t)nsteps * 4 * RED_SECTION * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->stop_host, c->stop_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    const int stop_step = *c->stop_host;
    *done = stop_step ? stop_step : k1;
    return LK_OK;
}

// One Lanczos step on the host-synchronous schedule (bases beyond KMAX_WIDE columns): lanczos.fypp:25-39, 57-62 through the ABI's own entries.
static int lanczos_step_sync(lk_linop_t A, lk_basis_t X, double *T, int64_t ldt, int k, double tol, int *info, int *stop) {
    const int ED = X->ed();
    *stop = 0;
    LKCHK(lk_linop_apply(A, LK_OP_N, X, k - 1, X, k));                                       // :26
    double *Tk = T + (size_t)(k - 1) * ldt * ED;
    for (int i = (k > 1 ? k - 1 : 1); i <= k; ++i) {                                         // :57-60
        double t[2] = {0.0, 0.0};
        LKCHK(lk_vec_dot(X, i - 1, X, k, t));
        for (int e = 0; e < ED; ++e) Tk[(size_t)(i - 1) * ED + e] = t[e];
        const double mt[2] = {-t[0], -t[1]}, one[2] = {1.0, 0.0};
        LKCHK(lk_vec_axpby(mt, X, i - 1, one, X, k));
    }
    int dinfo = 0;
    double norms[3];
    LKCHK(lk_dgs(X, k, X, k, nullptr, norms, 0, &dinfo));                                    // :62 (no beta)
    const double beta = norms[2];
    if (beta != beta) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
    Tk[(size_t)k * ED] = beta;                                                               // :29
    if (ED == 2) Tk[(size_t)k * ED + 1] = 0.0;
    if (beta < tol) { *info = k; *stop = 1; return LK_OK; }                                  // :32-36 (no scaling)
    const double inv[2] = {1.0 / beta, 0.0};
    return lk_vec_scal(X, k, inv);                                                           // :39
}

int lk_lanczos(lk_linop_t A, lk_basis_t X, double *T, int64_t ldt, int kstart, int kend, double tol, int *info) {
    if (!A || !X || !T || !info) return fail(LK_ERR_INVALID, "lk_lanczos: null argument");
    const int kdim = X->ncols - 1;                               // lanczos.fypp:20
    if (kdim < 1) return fail(LK_ERR_INVALID, "lk_lanczos: basis needs at least 2 columns");
    if (kstart < 1 || kend > kdim || kstart > kend + 1) return fail(LK_ERR_INVALID, "lk_lanczos: bad kstart/kend %d..%d (kdim %d)", kstart, kend, kdim);
    if (ldt < kdim + 1) return fail(LK_ERR_INVALID, "lk_lanczos: ldt too small");
    lk_context_t c = X->ctx;
    DevGuard dev_guard(c);
    const int ED = X->ed();
    *info = 0;
    LKCHK(lazy_enter(c, true));
    int k = kstart;
    const int kfused = kend < KMAX_WIDE ? kend : KMAX_WIDE;          // the asynchronous batch holds up to KMAX_WIDE basis columns per sweep
    while (k <= kend) {
        if (k > KMAX_WIDE) {
            // beyond 512 basis columns (round 5; the reference has no cap, lanczos.fypp:20): one step at a time on the host-synchronous
            // schedule -- operator, the two local orthogonalisations (:57-60), the full re-orthogonalisation on column panels (:62), norm, scale
            int stop = 0;
            LKCHK(lanczos_step_sync(A, X, T, ldt, k, tol, info, &stop));
            if (stop) break;
            ++k;
            continue;
        }
        int done = 0;
        resident_maybe_rearm(c);
        LKCHK(lanczos_batch_async(A, X, k, kfused, tol, &done));
        const bool stopped_early = *c->stop_host != 0;
        double beta = 0.0;
        const int rs = red_stride(kfused);
        int redo = 0;
        LKCHK(resident_status(X, done, c->step_red_host + (size_t)(done - k) * RED_SECTIONS * rs, rs, &redo));
        const int upto = redo ? done - 1 : done;              // (a single launch that gave up stopped the batch at `done`: that step runs again)
        for (int s = k; s <= upto; ++s) {
            const double *r2 = c->step_red_host + ((size_t)(s - k) * RED_SECTIONS + 2) * rs;
            const int i0 = s > 1 ? s - 1 : 1;
            double *Ts = T + (size_t)(s - 1) * ldt * ED;
            for (int i = i0; i <= s; ++i) {
                const double *a = c->lz_red_host + ((size_t)(s - k) * 4 + 2 * (i - i0)) * RED_SECTION;
                for (int e = 0; e < ED; ++e) Ts[(size_t)(i - 1) * ED + e] = a[e];           // T(i, k)   :58
            }
            beta = std::sqrt(std::fabs(r2[(size_t)s * ED]));
            if (beta != beta) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
            Ts[(size_t)s * ED] = beta;                                                      // T(k+1, k) :29
            if (ED == 2) Ts[(size_t)s * ED + 1] = 0.0;
        }
        if (redo) { k = done; continue; }
        if (!stopped_early) { k = done + 1; continue; }                                       // (on to the steps beyond KMAX_WIDE, if any)
        if (beta < tol) { *info = done; break; }                                            // :32-36 (no scaling)
        // the device stops at max(tol, atol_dp); a caller's smaller tol lets the reference go on: normalise and resume
        const double inv[2] = {1.0 / beta, 0.0};
        LKCHK(lk_vec_scal(X, done, inv));
        k = done + 1;
    }
    return LK_OK;
}

// Golub-Kahan steps [k0, k1] enqueued back to back (golub_kahan.fypp:25-60).  A step has two halves, each ending in a
// normalise kernel that can raise the stop flag, so the guard counts HALF steps: 2k - 1 = right half (V(k) = A^H U(k), DGS
// against V(:k-1), alpha), 2k = left half (U(k+1) = A V(k), DGS against U(:k), beta).  Half-step slots in step_red.
static int bidiag_batch_async(lk_linop_t A, lk_basis_t U, lk_basis_t V, int k0, int k1, double tol, int *done_half) {
    lk_context_t c = U->ctx;
    const int nsteps = k1 - k0 + 1;
    const int rs = red_stride(k1);
    LKCHK(ensure_step_buffers(c, 2 * nsteps, rs));
    HIPCHK(hipMemsetAsync(c->stop_dev, 0, sizeof(int), c->stream));
    const double tol_break = tol > ATOL_DP ? tol : ATOL_DP;
    c->guard_on = true;
    c->prof_sweeps_only = true;
    int rc = LK_OK;
    for (int k = k0; k <= k1 && rc == LK_OK; ++k) {
        double *sv = c->step_red + (size_t)(2 * (k - k0)) * RED_SECTIONS * rs;
        double *su = sv + (size_t)RED_SECTIONS * rs;
        c->guard_step = 2 * k - 1;
        rc = lk_linop_apply(A, LK_OP_H, U, k - 1, V, k - 1);                                   // :27
        if (rc != LK_OK) break;
        if (k > 1) rc = dgs_step_async(V, k - 1, V, k - 1, sv, rs, tol_break, tol_break);     // :30-33, :36-42
        else {
            rc = dot_device(V, 0, V, 0, sv + 2 * rs);                                         // ||V(1)||^2 where the DGS would leave it
            if (rc == LK_OK) rc = scal_launch(V, 0, 1.0, 0.0, sv + 2 * rs, tol_break, c->stop_dev, tol_break);
        }
        if (rc != LK_OK) break;
        c->guard_step = 2 * k;
        rc = lk_linop_apply(A, LK_OP_N, V, k - 1, U, k);                                       // :45
        if (rc != LK_OK) break;
        rc = dgs_step_async(U, k, U, k, su, rs, tol_break, tol_break);                        // :48-49, :52-58
    }
    c->guard_on = false;
    c->guard_step = 0;
    c->prof_sweeps_only = false;
    LKCHK(rc);
    HIPCHK(hipMemcpyAsync(c->step_red_host, c->step_red, (size_t)2 * nsteps * RED_SECTIONS * rs * sizeof(double),
                          hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->stop_host, c->stop_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    const int stop_half = *c->stop_host;
    *done_half = stop_half ? stop_half : 2 * k1;
    return LK_OK;
}

int lk_bidiag(lk_linop_t A, lk_basis_t U, lk_basis_t V, double *B, int64_t ldb, int kstart, int kend, double tol, int *info) {
    if (!A || !U || !V || !B || !info) return fail(LK_ERR_INVALID, "lk_bidiag: null argument");
    LKCHK(check_pair(U, V, "lk_bidiag"));
    const int kdim = U->ncols - 1;                               // golub_kahan.fypp:18
    if (kdim < 1 || V->ncols < kdim) return fail(LK_ERR_INVALID, "lk_bidiag: U needs kdim + 1 columns and V kdim");
    if (kstart < 1 || kend > kdim || kstart > kend + 1) return fail(LK_ERR_INVALID, "lk_bidiag: bad kstart/kend %d..%d (kdim %d)", kstart, kend, kdim);
    if (!(tol >= ATOL_DP)) return fail(LK_ERR_INVALID, "lk_bidiag: tol below atol_dp is not fused (the device-side stop is at max(tol, atol_dp))");
    if (ldb < kdim + 1) return fail(LK_ERR_INVALID, "lk_bidiag: ldb too small");
    if (U->data == V->data) return fail(LK_ERR_INVALID, "lk_bidiag: U and V must be different bases");
    lk_context_t c = U->ctx;
    DevGuard dev_guard(c);
    const int ED = U->ed();
    *info = 0;
    if (kstart > kend) return LK_OK;
    LKCHK(lazy_enter(c, true));
    const int kfused = kend < KMAX_WIDE ? kend : KMAX_WIDE;
    int done_half = 2 * (kstart - 1);
    bool stopped_early = false;
    for (int ks = kstart; ks <= kfused;) {
        resident_maybe_rearm(c);
        LKCHK(bidiag_batch_async(A, U, V, ks, kfused, tol, &done_half));
        stopped_early = *c->stop_host != 0;
        const int rs = red_stride(kfused);
        int redo = 0;
        if (stopped_early) {
            // the half step the batch stopped at: did its single-launch Gram-Schmidt step give up?  Then the whole step runs again
            // (both halves: recomputing the right half from the untouched U(k) gives the same bits) on the three-sweep schedule.
            const int kq = (done_half + 1) / 2;
            const bool right = done_half & 1;
            const double *slot = c->step_red_host + (size_t)(done_half - (2 * ks - 1)) * RED_SECTIONS * rs;
            if (!(right && kq == 1)) LKCHK(resident_status(right ? V : U, right ? kq - 1 : kq, slot, rs, &redo));
        }
        const int upto = redo ? done_half - 1 : done_half;
        for (int hs = 2 * ks - 1; hs <= upto; ++hs) {
            const int k = (hs + 1) / 2;
            const bool right = hs & 1;
            const double *r2 = c->step_red_host + ((size_t)(hs - (2 * ks - 1)) * RED_SECTIONS + 2) * rs;
            const double nrm = std::sqrt(std::fabs(r2[(size_t)(right ? k - 1 : k) * ED]));
            if (nrm != nrm) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
            double *Bk = B + (size_t)(k - 1) * ldb * ED;
            const size_t row = right ? (size_t)(k - 1) : (size_t)k;                           // B(k, k) = alpha ; B(k+1, k) = beta
            Bk[row * ED] = nrm;
            if (ED == 2) Bk[row * ED + 1] = 0.0;
        }
        if (!redo) break;
        ks = (done_half + 1) / 2;
        stopped_early = false;
    }
    if (stopped_early) { *info = (done_half + 1) / 2; return LK_OK; }                     // :41, :57
    // beyond 512 basis columns (round 5; the reference has no cap, golub_kahan.fypp:18): one step at a time, host-synchronous
    for (int k = (kstart > kfused + 1 ? kstart : kfused + 1); k <= kend; ++k) {
        double *Bk = B + (size_t)(k - 1) * ldb * ED;
        double norms[3];
        int dinfo = 0;
        LKCHK(lk_linop_apply(A, LK_OP_H, U, k - 1, V, k - 1));                             // :27
        if (k > 1) LKCHK(lk_dgs(V, k - 1, V, k - 1, nullptr, norms, 0, &dinfo));           // :30-33
        else LKCHK(lk_vec_norm(V, 0, &norms[2]));
        const double alpha = norms[2];                                                     // :36
        if (alpha != alpha) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
        Bk[(size_t)(k - 1) * ED] = alpha;
        if (ED == 2) Bk[(size_t)(k - 1) * ED + 1] = 0.0;
        if (!(std::fabs(alpha) > tol)) { *info = k; return LK_OK; }                        // :37-42
        const double ia[2] = {1.0 / alpha, 0.0};
        LKCHK(lk_vec_scal(V, k - 1, ia));
        LKCHK(lk_linop_apply(A, LK_OP_N, V, k - 1, U, k));                                 // :45
        LKCHK(lk_dgs(U, k, U, k, nullptr, norms, 0, &dinfo));                              // :48-49
        const double beta = norms[2];                                                      // :52
        if (beta != beta) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
        Bk[(size_t)k * ED] = beta;
        if (ED == 2) Bk[(size_t)k * ED + 1] = 0.0;
        if (!(std::fabs(beta) > tol)) { *info = k; return LK_OK; }                         // :53-58
        const double ib[2] = {1.0 / beta, 0.0};
        LKCHK(lk_vec_scal(U, k, ib));
    }
    return LK_OK;
}

static int arnoldi_impl(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int kstart, int kend, double tol, int trans, int *info,
                        const int *seg_last, int nseg, lk_progress_fn fn, void *user) {
    if (!A || !X || !H || !info) return fail(LK_ERR_INVALID, "lk_arnoldi: null argument");
    const int kdim = X->ncols - 1;                               // arnoldi.fypp:26 (p = 1)
    if (kdim < 1) return fail(LK_ERR_INVALID, "lk_arnoldi: basis needs at least 2 columns");
    if (kstart < 1 || kend > kdim || kstart > kend + 1) return fail(LK_ERR_INVALID, "lk_arnoldi: bad kstart/kend %d..%d (kdim %d)", kstart, kend, kdim);
    if (ldh < kdim + 1) return fail(LK_ERR_INVALID, "lk_arnoldi: ldh too small");
    for (int i = 0; i < nseg; ++i)
        if (seg_last[i] < kstart || seg_last[i] > kend || (i > 0 && seg_last[i] <= seg_last[i - 1]))
            return fail(LK_ERR_INVALID, "lk_arnoldi_segments: segment ends must ascend within [kstart, kend]");
    lk_context_t c = X->ctx;
    DevGuard dev_guard(c);
    const int ED = X->ed();
    *info = 0;
    std::vector<double> h((size_t)kdim * ED);
    int k = kstart;
    int reported = kstart - 1;                                   // last step handed to `fn`
    bool stop_requested = false;
    auto report = [&](int upto) {
        if (fn && upto > reported) {
            if (fn(user, reported + 1, upto) != 0) stop_requested = true;
            reported = upto;
        }
    };
    while (k <= kend) {
        int stop = 0;
        if (!c->async_arnoldi || k > KMAX_WIDE || k == kend) {
            // single step, a basis beyond KMAX_WIDE columns, or the round-1 schedule: one host round trip per step
            LKCHK(arnoldi_step_sync(A, X, H, ldh, k, tol, trans, h, info, &stop));
            report(k);
            if (stop || stop_requested) break;
            ++k;
            continue;
        }
        const int k1 = kend < KMAX_WIDE ? kend : KMAX_WIDE;
        const int rs = red_stride(k1);
        LKCHK(lazy_enter(c, true));
        const int kb = k;                                        // first step of this batch (slot 0)
        // columns [sa, sb] of H from the batch's result slots; *stop_out = 1 when the reference's loop exits at a step (info set)
        int redo = 0;                                            // step whose single-launch Gram-Schmidt gave up (the batch stopped there)
        auto fill = [&](int sa, int sb, int *stop_out) -> int {
            for (int s = sa; s <= sb; ++s) {
                const double *slot = c->step_red_host + (size_t)(s - kb) * RED_SECTIONS * rs;
                const double *r0 = slot, *r1 = slot + rs, *r2 = slot + 2 * rs;
                if (resident_applies(X, s)) {
                    const double status = r2[(size_t)s * ED + 1];
                    if (status == 1.0) {                          // y untouched: this step runs again on the three-sweep schedule
                        LKCHK(resident_recover(c));
                        redo = s;
                        return LK_OK;
                    }
                    if (status != 0.0) return fail(LK_ERR_HIP, "lk_arnoldi: the single-launch Gram-Schmidt step failed after its first phase (status %g)", status);
                }
                double *Hk = H + (size_t)(s - 1) * ldh * ED;
                for (int i = 0; i < s * ED; ++i) Hk[i] = r0[i] + r1[i];                 // gram_schmidt.fypp:49
                const double beta = std::sqrt(std::fabs(r2[s * ED]));
                if (beta != beta) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
                Hk[(size_t)s * ED] = 0.0;
                if (ED == 2) Hk[(size_t)s * ED + 1] = 0.0;
                if (beta < ATOL_DP) {
                    // colinear column (the device left y'' unnormalised and stopped): R(1,1) = 0, rand, re-normalise  qr.fypp:146-162
                    LKCHK(lk_vec_rand(X, s, 0x5EEDull + (uint64_t)s, c->row0, 1));
                } else {
                    Hk[(size_t)s * ED] = beta;
                }
                if (std::fabs(Hk[(size_t)s * ED]) < tol) {                              // arnoldi.fypp:58-71
                    *info = s;
                    *stop_out = 1;
                    return LK_OK;
                }
            }
            return LK_OK;
        };
        // segments of THIS batch: the caller's boundaries that fall inside (k, k1)
        std::vector<int> segs;
        for (int i = 0; i < nseg; ++i)
            if (seg_last[i] >= k && seg_last[i] < k1) segs.push_back(seg_last[i]);
        auto deliver = [&](int sa, int sb) -> int {              // (only steps the device finished without raising the stop flag)
            int st = 0;
            LKCHK(fill(sa, sb, &st));
            if (st) return fail(LK_ERR_INVALID, "internal: a delivered segment stopped");   // cannot happen: tol_break >= tol on the device
            report(sb);
            return stop_requested ? LK_STOP_REQUESTED : LK_OK;
        };
        int done = 0, delivered_to = k - 1;
        bool cancelled = false;
        resident_maybe_rearm(c);
        LKCHK(arnoldi_batch_async(A, X, k, k1, tol, trans, &done, segs.empty() ? nullptr : segs.data(), (int)segs.size(), deliver, &delivered_to,
                                  &cancelled));
        if (cancelled) break;                                    // the caller asked to stop: columns beyond the last report are not delivered
        LKCHK(fill(delivered_to + 1, done, &stop));
        if (redo) {                                              // (the device stop flag was raised by the launch that gave up)
            report(redo - 1);
            if (stop_requested) break;
            k = redo;
            continue;
        }
        report(stop ? *info : done);
        if (stop || stop_requested) break;
        // a stop the reference would NOT have taken (tol below atol_dp with a colinear column): resume after it
        k = done + 1;
    }
    return LK_OK;
}

// ---- qr_no_pivoting and the block Arnoldi factorisation ---------------------------------------------------------------------------
// a view of columns [c0, c0 + ncols) of a panel (not owned): the basis Q(:j-1) a column of the block is orthogonalised against
static lk_basis_s basis_view(lk_basis_t B, int c0, int ncols) {
    lk_basis_s v = *B;
    v.data = B->col(c0);
    v.ncols = ncols;
    v.own = false;
    v.hwm = ncols;
    return v;
}

constexpr uint64_t QR_SEED = 0x5EEDull;       // the colinear-column re-draw of column c of a panel uses the counter stream QR_SEED + c + 1 (lk_arnoldi: + k)

// Columns [jbeg, p) of the block that starts at column c0 of X: qr_no_pivoting's loop body (qr.fypp:129-165), host-synchronous.
// `have_first`: column jbeg has been orthogonalised and its norm is beta0 (an asynchronous batch stopped right behind it; the device
// scaled it iff beta0 >= atol_dp).  R: p x p column-major (leading dimension ldr elements), only columns >= jbeg are written.
static int qr_columns_sync(lk_basis_t X, int c0, int p, int jbeg, bool have_first, double beta0, double *R, int64_t ldr, double tol, bool *flag,
                           int *info) {
    lk_context_t c = X->ctx;
    const int ED = X->ed();
    std::vector<double> hcol((size_t)(p > 1 ? p : 1) * ED);
    for (int j = jbeg; j < p; ++j) {
        double *Rj = R + (size_t)j * ldr * ED;
        double beta;
        bool scaled = false;
        if (have_first && j == jbeg) {
            beta = beta0;
            scaled = beta0 >= ATOL_DP;
        } else {
            for (int i = 0; i < p * ED; ++i) Rj[i] = 0.0;                                    // R = zero   :125
            if (j > 0) {
                lk_basis_s Qv = basis_view(X, c0, j);
                double norms[3];
                int dinfo = 0;
                LKCHK(lk_dgs(&Qv, j, X, c0 + j, hcol.data(), norms, 0, &dinfo));           // :131-134
                memcpy(Rj, hcol.data(), (size_t)j * ED * sizeof(double));
                beta = norms[2];                                                           // :135 (the norm of what the step left)
            } else {
                LKCHK(lk_vec_norm(X, c0, &beta));
            }
        }
        if (beta != beta) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");           // :137-143
        if (std::fabs(beta) < tol) {                                                       // colinear column  :146-162
            if (!*flag) { *flag = true; *info = j + 1; }
            Rj[(size_t)j * ED] = 0.0;
            if (ED == 2) Rj[(size_t)j * ED + 1] = 0.0;
            LKCHK(lk_vec_rand(X, c0 + j, QR_SEED + (uint64_t)(c0 + j) + 1, c->row0, 0));
            if (j > 0) {
                lk_basis_s Qv = basis_view(X, c0, j);
                int dinfo = 0;
                LKCHK(lk_dgs(&Qv, j, X, c0 + j, nullptr, nullptr, 0, &dinfo));
            }
            LKCHK(lk_vec_norm(X, c0 + j, &beta));
            scaled = false;
        } else {
            Rj[(size_t)j * ED] = beta;
            if (ED == 2) Rj[(size_t)j * ED + 1] = 0.0;
        }
        if (!scaled) {
            const double inv[2] = {1.0 / beta, 0.0};
            LKCHK(lk_vec_scal(X, c0 + j, inv));                                             // :164
        }
    }
    return LK_OK;
}

int lk_qr(lk_basis_t Q, int j0, int p, double *R, int64_t ldr, double tol, int *info) {
    if (!Q || !R || !info) return fail(LK_ERR_INVALID, "lk_qr: null argument");
    if (p < 1 || j0 < 0 || j0 + p > Q->ncols) return fail(LK_ERR_INVALID, "lk_qr: bad column range");
    if (ldr < p) return fail(LK_ERR_INVALID, "lk_qr: ldr too small");
    DevGuard dev_guard(Q->ctx);
    LKCHK(lazy_enter(Q->ctx, true));
    *info = 0;
    bool flag = false;
    return qr_columns_sync(Q, j0, p, 0, false, 0.0, R, ldr, tol, &flag, info);
}

// Block Arnoldi steps [k0, k1] enqueued back to back (arnoldi.fypp:34-73 with blksize = p): p operator applications, the panel x panel
// Gram-Schmidt of the new block against X(:, :kp), and qr_no_pivoting of the block column by column -- column j's single-vector step
// against the j columns before it (the single launch of lk_resident.hip.h for cache-resident panels), its norm and scale with the device
// stop flag.  The guard counts (step, column): seq = (k - k0) (p + 1) + 1 for the operator + block part, + 1 + j for column j, so a
// column below max(tol, atol_dp) -- or a single launch that gave up -- stops everything behind it, the rest of that block included
// (the host finishes it: qr_columns_sync).  One copy + one synchronisation per batch.
static int arnoldi_block_batch(lk_linop_t A, lk_basis_t X, int p, int k0, int k1, double tol, int trans, std::vector<int64_t> &blk_off,
                               int64_t *qr_off) {
    lk_context_t c = X->ctx;
    const int ED = X->ed();
    const int nsteps = k1 - k0 + 1;
    const int rs = RED_SECTION;                                          // the block's columns see at most p - 1 <= 31 basis columns
    blk_off.assign(nsteps + 1, 0);
    for (int s = 0; s < nsteps; ++s) blk_off[s + 1] = blk_off[s] + dgs_block_slot_doubles(X, (k0 + s) * p, p);
    *qr_off = blk_off[nsteps];
    const int64_t total = *qr_off + (int64_t)nsteps * p * RED_SECTIONS * rs;
    LKCHK(ensure_block_buffers(c, total));
    HIPCHK(hipMemsetAsync(c->stop_dev, 0, sizeof(int), c->stream));
    const double tol_break = tol > ATOL_DP ? tol : ATOL_DP;
    c->guard_on = true;
    int rc = LK_OK;
    for (int k = k0; k <= k1 && rc == LK_OK; ++k) {
        const int kpm = (k - 1) * p, kp = k * p;
        const int seq0 = (k - k0) * (p + 1) + 1;
        c->guard_step = seq0;
        for (int i = 0; i < p && rc == LK_OK; ++i) rc = lk_linop_apply(A, trans ? LK_OP_H : LK_OP_N, X, kpm + i, X, kp + i);   // :39-47
        if (rc != LK_OK) break;
        rc = dgs_block_walk(X, kp, X, kp, p, c->blk_red + blk_off[k - k0], nullptr, nullptr, nullptr, 0);                 // :50-51
        for (int j = 0; j < p && rc == LK_OK; ++j) {                                                                    // :55
            c->guard_step = seq0 + 1 + j;
            double *slot = c->blk_red + *qr_off + ((int64_t)(k - k0) * p + j) * RED_SECTIONS * rs;
            if (j > 0) {
                lk_basis_s Qv = basis_view(X, kp, j);
                rc = dgs_step_async(&Qv, j, X, kp + j, slot, rs, ATOL_DP, tol_break);
            } else {
                rc = dot_device(X, kp, X, kp, slot + 2 * rs);
                if (rc == LK_OK) rc = scal_launch(X, kp, 1.0, 0.0, slot + 2 * rs, ATOL_DP, c->stop_dev, tol_break);
            }
        }
    }
    c->guard_on = false;
    c->guard_step = 0;
    LKCHK(rc);
    HIPCHK(hipMemcpyAsync(c->blk_red_host, c->blk_red, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->stop_host, c->stop_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->prof) prof_collect(c);
    (void)ED;
    return LK_OK;
}

int lk_arnoldi_block(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int blksize, int kstart, int kend, double tol, int trans, int *info) {
    if (!A || !X || !H || !info) return fail(LK_ERR_INVALID, "lk_arnoldi_block: null argument");
    const int p = blksize;
    if (p < 1) return fail(LK_ERR_INVALID, "lk_arnoldi_block: blksize must be positive");
    if (p == 1) return lk_arnoldi(A, X, H, ldh, kstart, kend, tol, trans, info);
    const int kdim = (X->ncols - p) / p;                          // arnoldi.fypp:26
    if (kdim < 1) return fail(LK_ERR_INVALID, "lk_arnoldi_block: the basis needs at least 2 blocks of %d columns", p);
    if (kstart < 1 || kend > kdim || kstart > kend + 1) return fail(LK_ERR_INVALID, "lk_arnoldi_block: bad kstart/kend %d..%d (kdim %d)", kstart, kend, kdim);
    if (ldh < (int64_t)(kdim + 1) * p) return fail(LK_ERR_INVALID, "lk_arnoldi_block: ldh too small");
    lk_context_t c = X->ctx;
    DevGuard dev_guard(c);
    const int ED = X->ed();
    *info = 0;
    LKCHK(lazy_enter(c, true));
    auto Hat = [&](int i, int j) { return H + ((size_t)j * ldh + i) * ED; };
    // one step on the host-synchronous schedule (wide bases, "async_arnoldi" = 0): the entries of the ABI, one round trip each
    auto step_sync = [&](int k, int *stop) -> int {
        const int kpm = (k - 1) * p, kp = k * p;
        for (int i = 0; i < p; ++i) LKCHK(lk_linop_apply(A, trans ? LK_OP_H : LK_OP_N, X, kpm + i, X, kp + i));
        std::vector<double> hb((size_t)kp * p * ED), R((size_t)p * p * ED, 0.0);
        int dinfo = 0;
        lk_basis_s Yv = basis_view(X, kp, p);
        LKCHK(lk_dgs_block(X, kp, &Yv, 0, p, hb.data(), &dinfo));
        for (int q = 0; q < p; ++q) memcpy(Hat(0, kpm + q), hb.data() + (size_t)q * kp * ED, (size_t)kp * ED * sizeof(double));
        int qinfo = 0;
        bool flag = false;
        LKCHK(qr_columns_sync(X, kp, p, 0, false, 0.0, R.data(), p, ATOL_DP, &flag, &qinfo));
        for (int q = 0; q < p; ++q) memcpy(Hat(kp, kpm + q), R.data() + (size_t)q * p * ED, (size_t)p * ED * sizeof(double));
        double mn = HUGE_VAL;
        for (int i = 0; i < p; ++i) mn = std::fmin(mn, std::fabs(Hat(kp + i, kpm + i)[0]));                            // :58-62 (real part)
        *stop = mn < tol;
        return LK_OK;
    };
    int k = kstart;
    while (k <= kend) {
        int stop = 0;
        const bool wide = (int64_t)(kend + 1) * p > KMAX_WIDE || !dgs_block_fused_ok(X, k * p, X, k * p, p);
        if (!c->async_arnoldi || wide || k == kend) {
            LKCHK(step_sync(k, &stop));
            if (stop) { *info = k * p; break; }                                                                       // :65-71
            ++k;
            continue;
        }
        std::vector<int64_t> blk_off;
        int64_t qr_off = 0;
        const int k0 = k;
        resident_maybe_rearm(c);
        LKCHK(arnoldi_block_batch(A, X, p, k0, kend, tol, trans, blk_off, &qr_off));
        const int rs = RED_SECTION;
        const int stop_seq = *c->stop_host;
        // (step, column) the batch stopped at: everything in front of it ran
        const int ks = stop_seq ? k0 + (stop_seq - 1) / (p + 1) : kend + 1;
        const int js = stop_seq ? (stop_seq - 1) % (p + 1) - 1 : -1;
        if (stop_seq && js < 0) return fail(LK_ERR_INVALID, "internal: the block Arnoldi batch stopped outside a column");
        const double *host = c->blk_red_host;
        bool exit_loop = false;
        for (int s = k0; s <= (stop_seq ? ks : kend); ++s) {
            const int kpm = (s - 1) * p, kp = s * p;
            std::vector<double> hb((size_t)kp * p * ED);
            int dinfo = 0;
            LKCHK(dgs_block_walk(X, kp, X, kp, p, nullptr, host + blk_off[s - k0], hb.data(), &dinfo, 1));
            for (int q = 0; q < p; ++q) memcpy(Hat(0, kpm + q), hb.data() + (size_t)q * kp * ED, (size_t)kp * ED * sizeof(double));
            const int jlast = (stop_seq && s == ks) ? js : p - 1;           // columns of this block the device finished (jlast: up to its norm)
            double beta_last = 0.0;
            bool redo_last = false;
            for (int j = 0; j <= jlast; ++j) {
                const double *slot = host + qr_off + ((int64_t)(s - k0) * p + j) * RED_SECTIONS * rs;
                double *Rj = Hat(kp, kpm + j);
                for (int i = 0; i < p * ED; ++i) Rj[i] = 0.0;
                if (j > 0) {
                    lk_basis_s Qv = basis_view(X, kp, j);
                    int redo = 0;
                    LKCHK(resident_status(&Qv, j, slot, rs, &redo));
                    if (redo) { redo_last = true; break; }                  // (only the column the batch stopped at can have given up)
                    for (int i = 0; i < j * ED; ++i) Rj[i] = slot[i] + slot[rs + i];                                  // gram_schmidt.fypp:49
                }
                const double beta = std::sqrt(std::fabs(slot[2 * rs + (size_t)j * ED]));
                if (beta != beta) return fail(LK_ERR_NAN, "|beta| = NaN detected! Abort");
                Rj[(size_t)j * ED] = beta;
                beta_last = beta;
            }
            if (stop_seq && s == ks) {
                // the host finishes the block behind the stop: the column that raised it (colinear: re-draw; below the caller's tol: already
                // scaled; a single launch that gave up: from scratch) and the columns after it
                bool flag = false;
                int qinfo = 0;
                std::vector<double> R((size_t)p * p * ED, 0.0);
                for (int q = 0; q < p; ++q) memcpy(R.data() + (size_t)q * p * ED, Hat(kp, kpm + q), (size_t)p * ED * sizeof(double));
                LKCHK(qr_columns_sync(X, kp, p, js, !redo_last, beta_last, R.data(), p, ATOL_DP, &flag, &qinfo));
                for (int q = js; q < p; ++q) memcpy(Hat(kp, kpm + q), R.data() + (size_t)q * p * ED, (size_t)p * ED * sizeof(double));
            }
            double mn = HUGE_VAL;
            for (int i = 0; i < p; ++i) mn = std::fmin(mn, std::fabs(Hat(kp + i, kpm + i)[0]));                        // :58-62
            if (mn < tol) { *info = kp; exit_loop = true; break; }                                                    // :65-71
        }
        if (exit_loop) break;
        k = stop_seq ? ks + 1 : kend + 1;                                   // (a stop the reference would not have taken: on with the next step)
    }
    return LK_OK;
}

int lk_arnoldi(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int kstart, int kend, double tol, int trans, int *info) {
    return arnoldi_impl(A, X, H, ldh, kstart, kend, tol, trans, info, nullptr, 0, nullptr, nullptr);
}

int lk_arnoldi_segments(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int kstart, int kend, double tol, int trans, const int *seg_last,
                        int nseg, lk_progress_fn fn, void *user, int *info) {
    if (nseg < 0 || (nseg > 0 && !seg_last)) return fail(LK_ERR_INVALID, "lk_arnoldi_segments: bad segment list");
    return arnoldi_impl(A, X, H, ldh, kstart, kend, tol, trans, info, seg_last, nseg, fn, user);
}

}  // extern "C"
