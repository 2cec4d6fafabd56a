/* lightkrylov_hip.h -- C ABI of the MI355X-native Krylov inner-loop engine.
 *
 * This is the drop-in boundary for LightKrylov's hot path: everything the reference's
 * Krylov layer (src/Krylov, src/IterativeSolvers) asks of a vector type goes through the
 * six deferred type-bound procedures of `abstract_vector_{rdp,cdp}` and the helpers built
 * on them.  Each entry point below names the reference interface it replaces
 * (file:line under the LightKrylov source tree).  Plain pointers and sizes only: a
 * Fortran `bind(C)` interface block (fortran/lk_hip_iso_c.f90), a ctypes stub
 * (lightkrylov_amd/_capi.py) or C code can bind it as is.
 *
 * Model
 *   - one context per process = one GPU (HIP device + stream).  Multi-GPU = one process
 *     per GPU, each holding a contiguous ROW BLOCK of every vector; the only coupling is
 *     the sum-reduction of dot/norm/projection coefficients, delivered through a
 *     user-installed all-reduce callback (RCCL in production).
 *   - a Krylov basis is ONE column-contiguous panel in HBM: element (i, j) lives at
 *     data[j * ld + i]; ld is padded so every column starts 256-byte aligned.
 *     A single vector is a 1-column basis; every vector argument is a (basis, column) pair
 *     with 0-based column indices.
 *   - dtype LK_F64  = real(dp)     (reference kind "rdp")
 *           LK_C128 = complex(dp)  (reference kind "cdp"), interleaved (re, im).
 *     Scalars cross the ABI as `const double*` to 1 (F64) or 2 (C128) doubles.
 *   - every call returns LK_OK (0) or a negative error code; lk_last_error() gives the text
 *     (the Fortran shim maps non-zero to LightKrylov's stop_error,
 *     src/Utilities/Logger.f90:290-298).  `info` out-arguments follow the reference
 *     convention (src/Krylov/BaseKrylov.fypp:106-109): 0 ok, >0 informational, <0 failure.
 *   - calls are asynchronous on the context's stream except those that return host
 *     scalars, which synchronise that stream once.
 *   There is NO CPU fallback: without a HIP device lk_init fails.
 */
#ifndef LIGHTKRYLOV_HIP_H
#define LIGHTKRYLOV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lk_context_s *lk_context_t;
typedef struct lk_basis_s *lk_basis_t;
typedef struct lk_linop_s *lk_linop_t;

enum { LK_F64 = 0, LK_C128 = 1 };

enum {
    LK_OK = 0,
    LK_ERR_INVALID = -1, /* bad argument / shape / dtype mismatch (reference: stop_error) */
    LK_ERR_HIP = -2,     /* HIP runtime error */
    LK_ERR_NOMEM = -3,
    LK_ERR_COMM = -4,    /* all-reduce callback failed */
    LK_ERR_NAN = -5      /* |beta| = NaN detected (src/Krylov/qr.fypp:137-143) */
};

/* lk_dgs flags */
enum {
    LK_DGS_NORMALIZE = 1 /* fold qr_no_pivoting's 1-column normalise (qr.fypp:134,164) into the call:
                            y <- y / ||y|| on the device when ||y|| >= atol_dp */
};

/* matvec transposition selector for lk_linop_apply (AbstractLinops.fypp:391-424) */
enum { LK_OP_N = 0, LK_OP_H = 1 };

/* ---- context ------------------------------------------------------------------------- */

/* Sum all-reduce over the ranks that share a row-sharded vector: `count` doubles at device
 * address `dev_buf`, in place, ordered on `stream` (a hipStream_t).  Return 0 on success.
 * The reference has no collective at all (reductions are "the user's job inside dot",
 * paper/paper.md:35,97,101); this hook is where that job is done once per sweep. */
typedef int (*lk_allreduce_fn)(void *user, void *dev_buf, int64_t count, void *stream);

int lk_version(void);
const char *lk_last_error(void);

/* device: HIP device ordinal.  stream: a hipStream_t to run on, or NULL for a stream owned
 * by the context. */
int lk_init(int device, void *stream, lk_context_t *ctx);
int lk_finalize(lk_context_t ctx);
int lk_sync(lk_context_t ctx);
int lk_set_allreduce(lk_context_t ctx, lk_allreduce_fn fn, void *user, int nranks, int rank);
/* device ordinal and hipStream_t the context runs on (either pointer may be NULL). */
int lk_context_info(lk_context_t ctx, int *device, void **stream);
/* ranks sharing the row-sharded basis and this context's rank, as installed by lk_set_allreduce or lk_comm_init_rank (1, 0 without
 * either; either pointer may be NULL).  What a host-side check of a row partition must read -- not a shadow of its own. */
int lk_comm_info(lk_context_t ctx, int *nranks, int *rank);

/* Native RCCL all-reduce (one process per GPU; backend "nccl" of the reference-side launchers IS RCCL on
 * ROCm).  Rank 0 calls lk_comm_get_unique_id and ships the LK_COMM_ID_BYTES opaque bytes to every rank by
 * whatever out-of-band channel the host program has (MPI_Bcast in a Fortran/MPI host, the torch.distributed
 * store in bench.py); every rank then calls lk_comm_init_rank (collective: ncclCommInitRank on the context's
 * device), which installs ncclAllReduce(ncclDouble, ncclSum) on the context's stream as the engine's
 * reduction hook -- the <= 129 (258 complex) scalars of each sweep are reduced in place in device memory.
 * The reference itself has no collective (paper/paper.md:35,97,101: "the user's job inside dot").
 * librccl is dlopen'ed on first use.  lk_comm_destroy (or lk_finalize) releases the communicator.
 * lk_comm_available is the LOCAL half of that (no collective, no GPU work): LK_OK when librccl and every entry point the
 * communicator needs resolve in this process -- so that the ranks of a job can agree on the route (native, or a host-provided
 * lk_set_allreduce) BEFORE anyone enters the collective lk_comm_init_rank, where a rank that fails alone leaves the others
 * waiting in the bootstrap (bench.py: the flag is all-reduced over the launcher's process group). */
#define LK_COMM_ID_BYTES 128
int lk_comm_available(void);
int lk_comm_get_unique_id(void *id_out);
int lk_comm_init_rank(lk_context_t ctx, int nranks, int rank, const void *id);
int lk_comm_destroy(lk_context_t ctx);

/* Nearest-neighbour exchange between consecutive ranks of the row partition, needed only by stencil operators
 * (5-point Laplacian: one grid line; Ginzburg-Landau: one point): `count` doubles at device address send_lo go to
 * rank-1 and send_hi to rank+1; recv_lo is filled with rank-1's send_hi, recv_hi with rank+1's send_lo; a NULL pair
 * means "no neighbour on that side".  Ordered on `stream`.  lk_comm_init_rank installs the native one
 * (ncclSend / ncclRecv in one group); lk_set_halo_exchange lets a host bring its own (MPI_Sendrecv, tests). */
typedef int (*lk_halo_fn)(void *user, const void *send_lo, const void *send_hi, void *recv_lo, void *recv_hi,
                          int64_t count, void *stream);
int lk_set_halo_exchange(lk_context_t ctx, lk_halo_fn fn, void *user);

/* All-gather of row blocks, needed only by the row-sharded dense and CSR operators (their matvec needs the whole input
 * vector): rank r contributes counts[r] doubles -- this rank's at device address `send` --, and on return the device buffer
 * `recv` holds every rank's block, rank r's at recv + displs[r] (doubles).  Ordered on `stream`.  lk_comm_init_rank
 * installs the native one (ncclAllGather for equal blocks, else ncclSend / ncclRecv in one group); lk_set_allgather lets a host
 * bring its own (MPI_Allgatherv, tests). */
typedef int (*lk_allgather_fn)(void *user, const void *send, void *recv, const int64_t *counts, const int64_t *displs,
                               int nranks, void *stream);
int lk_set_allgather(lk_context_t ctx, lk_allgather_fn fn, void *user);

/* row block owned by this rank: global rows [row0, row0 + n_local) of n_global; only used
 * so that counter-based rand fills are identical for every partition. */
int lk_set_partition(lk_context_t ctx, int64_t row0, int64_t n_global);
/* Tuning keys (integers; 30 of them -- round 6 removed every key whose other setting was measured slower and never defaulted, the
 * record of those A/Bs is docs/TUNING_LOG.md).  None changes a result beyond rounding; the ones marked [bits] change no result bit.
 *   schedule      "async_arnoldi" (1: lk_arnoldi / lk_lanczos / lk_bidiag enqueue all steps behind a device-side breakdown flag, one host
 *                 synchronisation per call; 0: one round trip per step) [bits]; "lazy", "lazy_speculate" (see lk_lazy_stats) [bits];
 *                 "pool_slab_cols" (columns per pool slab) [bits]
 *   single launch "resident", "resident_max_mb", "resident_onchip", "resident_rev", "resident_spin_ms" (see lk_resident_stats)
 *   sweeps        "recompute_update" (1: sweep 2 keeps y' in registers, sweep 3 re-forms it: y' never goes to HBM) [bits]; "store_policy"
 *                 (cache policy of the sweeps' 16-byte y store: 0 plain, 1 nt, 2 sc1 = write-through [default], 3 sc0 sc1) and
 *                 "store_split" [bits]; "dot_colwise" (1: sweep 1 / innerprod one column at a time, panel_dot_cw; 0: all columns per
 *                 tile) with "cw_u" (16-byte loads per lane and column: 4, 8, 0 = by size) and "cw_grid_mult"; "grid_mult", "blas1_grid_mult" (blocks per CU: the number of per-block partial sums a dot is assembled from);
 *                 "wide_regs" (update sweeps against 129..384 columns: 2 = register tiles of 32 / 24 columns + the lane split on 24-column
 *                 groups, 1 = the first only, 0 = lane split on 16-column groups) and "wide_s3" (1: sweep 3 of a lane-split step holds both
 *                 column groups of a wave-column in one wave's registers) [bits]
 *   matrix cores  "xhy_mfma" (1: X^H Y with five or more right-hand sides -- Gram, innerprod_matrix, block Gram-Schmidt -- in one pass over
 *                 X on the FP64 MFMAs; 0: four right-hand sides per pass on the vector units); "block_fused" (block Gram-Schmidt: 1 = three
 *                 passes per group for the real kind, 2 = for both kinds, 0 = four); "gemm_mfma_min" (tall-skinny product on the MFMAs from
 *                 this many output columns on; default 5 real / 9 complex); "gemm_3m" (1: the complex MFMA kernels use three real products
 *                 per complex one); "xhy_db" (panel_xhy_mfma with a double-buffered LDS tile: 1 = the 128-column variants, 2 = all, 0 = never); "gemm_roll" (1: the real product with 33..64 outputs keeps a ring of four k-steps of X in flight; 2:
 *                 every variant that has a ring; 0: batches) [bits]; "gram_rs" (Gram matrix of 5..128 real columns by panel_gram_rs and of 5..112
 *                 complex columns by panel_gram_rs3m / panel_gram_rs3m4 -- rows of the staged tile dealt to the waves, tiles staged by LDS-DMA: 1 = the resident
 *                 number of blocks per CU, n > 1 = n blocks per CU, 0 = panel_xhy_mfma / panel_gram_mfma3m); "upd_rs" (fused pass of the block Gram-Schmidt, real kind, 17..32 right-hand sides: 1 = panel_xhy_upd_rs --
 *                 row-owner waves on LDS-DMA tiles, coefficients in registers --, 0 = panel_xhy_upd_mfma)
 *   operators     "csr_stream" (1: CSR product through LDS for matrices with short rows; 0: lanes-per-row kernel)
 * A build made with -DLK_DIAGNOSTICS (make -C lightkrylov_amd/csrc diagnostics; NOT what build() produces) adds "xhy_debug" / "upd_debug",
 * which switch parts of a kernel off for phase timing and give WRONG results; the shipped library rejects them as unknown keys. */
int lk_set_tuning(lk_context_t ctx, const char *key, int value);

/* Lazy batching of the per-object path (tuning key "lazy", off by default).  When on, k consecutive
 * lk_vec_dot(X, j, y) calls over the columns of one panel cost ONE sweep (the first call computes the whole
 * run, the rest are memo hits) and consecutive lk_vec_axpby(a_j, X, j, 1, y) calls are queued and applied as
 * ONE panel update when anything else touches the engine.  This is what turns the schedule an unchanged
 * LightKrylov drives through the type-bound procedures (innerprod / linear_combination loops,
 * AbstractVectors.fypp:672-674, 600-602) into fused traffic.  out4 = {dot memo hits, batched dot sweeps,
 * queued axpbys, queue flushes}. */
int lk_lazy_stats(lk_context_t ctx, int64_t *out4);
/* The temporary of linear_combination (AbstractVectors.fypp:595-603: `allocate(y, source=X(1)); y%zero(); y%axpby(..)`)
 * stays VIRTUAL in lazy mode: its zero() and its k axpbys are recorded, the `y%sub(proj)` that consumes it
 * (gram_schmidt.fypp:145) becomes a pending y -= X h, and the NEXT pass's `y%norm()` (gram_schmidt.fypp:126, qr.fypp:135)
 * runs one sweep that forms and stores y', and returns ||y'|| and X^H y' for the k dot calls that follow: ONE pass over
 * X per Gram-Schmidt pass, 3k+6 columns per double_gram_schmidt_step against 3k+4 for lk_dgs.  The temporary is
 * written only if something reads it.  out4 = {fused update+dot sweeps, pending updates applied as plain panel updates,
 * virtual temporaries dropped unwritten, virtual temporaries written after all}. */
int lk_lazy_fusion_stats(lk_context_t ctx, int64_t *out4);
/* The first pass of a Gram-Schmidt step opens with y%norm() and then asks X(1..k)%dot(y) (gram_schmidt.fypp:126, 141): a norm
 * kernel and the batched dot sweep, two host synchronisations -- although the sweep computes ||y||^2 on the side.  Once that pair
 * has been seen for y = column j of a panel, the norm of column j + 1 (the next Arnoldi / Lanczos step) runs the sweep at once and
 * serves the norm and the k dots from it (tuning key "lazy_speculate", default 1).  A prediction nobody uses disarms it.
 * out2 = {anticipated sweeps, of which unused}. */
int lk_lazy_speculation_stats(lk_context_t ctx, int64_t *out2);

/* Single-launch Gram-Schmidt step (round 6; csrc/lk_resident.hip.h).  When the panel X(:, :k) | y fits the 256 MB memory-side
 * cache -- the launch-bound regime of the reference's own use cases (1.75 10^5 unknowns, paper/paper.md:103-113) -- lk_dgs and every
 * step of lk_arnoldi run double_gram_schmidt_step (gram_schmidt.fypp:12-57) AND qr_no_pivoting's norm + scale (qr.fypp:135-165) as ONE
 * persistent kernel: each block keeps its rows for the three phases, the phases' sums meet inside the launch in a fixed order.  Panels
 * up to the register files' capacity (64 MB on the chip) stay IN REGISTERS for the whole step: X is read once, k + 2 columns of
 * traffic instead of 3k + 5.  Tuning keys: "resident" (default 1; 0 = always the three-sweep schedule), "resident_max_mb" (default 192:
 * MB of panel the single launch takes), "resident_onchip" (default 1; 0 = never the register-resident kernel), "resident_rev" (tile
 * order of the cache-resident kernel's phase 2) and "resident_spin_ms" (default 50: bound on the first grid-wide wait, which normally ends within microseconds -- 0 = give up there at
 * once, the tests' way to the fallback --; a launch that cannot get all its blocks on the chip, e.g. because another context's persistent
 * kernel holds part of it, gives up BEFORE writing anything, that step runs on the three-sweep schedule, and the single launch pauses for 16
 * steps -- doubling with every further give-up -- before it is tried again; setting "resident" = 1 re-arms it at once).  One rank only: a row-sharded context keeps the three launches, whose sums meet in the all-reduce.  Same
 * results to rounding (different summation order).
 * out3 = {single launches enqueued, launches that gave up, launches that kept the panel in registers}. */
int lk_resident_stats(lk_context_t ctx, int64_t *out3);
/* In-kernel timeline of the LAST single launch (profiling aid, like lk_profile_get): the 100 MHz wall clock of block 0 at
 * kernel entry | phase 1 done (the panel read included) | sum 1 done | phase 2 done | sum 2 done | phase 3 done | sum 3 done | scaled.  Synchronises the stream. */
int lk_resident_phase_ticks(lk_context_t ctx, int64_t *out8);

/* per-kernel HIP-event timing on the context's stream (bench.py roofline leg).
 * tags: "dgs_sweep1|2|3" (the three panel sweeps; "dgs_sweep*" sums them -- a trailing '*' is a
 * prefix match), "dgs" (whole lk_dgs call), "matvec", "blas1". */
int lk_profile_enable(lk_context_t ctx, int on);
int lk_profile_get(lk_context_t ctx, const char *tag, int64_t *count, double *total_ms,
                   double *total_bytes);
int lk_profile_reset(lk_context_t ctx);

/* ---- basis = array of vectors, X(:) in the reference --------------------------------- */

/* replaces `allocate(X(ncols), source=...)` + `zero_basis(X)` (AbstractVectors.fypp:711-715,
 * IterativeSolvers.fypp:1032-1034): n_local rows on this rank, columns zero-filled. */
int lk_basis_create(lk_context_t ctx, int dtype, int64_t n_local, int ncols, lk_basis_t *B);
/* wrap caller-owned device memory (e.g. a torch tensor).  ld in elements; dev_ptr 16-byte
 * aligned; ld even for LK_F64.  Rows [n_local, ld) are never read or written. */
int lk_basis_wrap(lk_context_t ctx, int dtype, int64_t n_local, int ncols, int64_t ld,
                  void *dev_ptr, lk_basis_t *B);
int lk_basis_destroy(lk_basis_t B);
/* shape and base address of the panel.  The address is for building views (lk_basis_wrap) and for memory the caller owns;
 * to READ OR WRITE a vector's contents from a kernel of your own use lk_vec_device_ptr, which first applies what the engine
 * may still owe that vector in lazy mode. */
int lk_basis_info(lk_basis_t B, int *dtype, int64_t *n_local, int *ncols, int64_t *ld,
                  void **dev_ptr);
/* A USER'S OWN KERNEL on a vector (a hand-written `matvec`, AbstractLinops.fypp:74-87): the device address of column j,
 * valid until the vector is released, with everything the engine still owes that vector applied first (lazy mode defers
 * updates).  `access` says what the caller will do: LK_ACCESS_READ, LK_ACCESS_OVERWRITE (previous contents not read:
 * `vec_out` is intent(out)) or LK_ACCESS_READWRITE.  The caller's work must be ordered after the engine's: enqueue it on
 * the context's stream (lk_context_info) or call lk_sync first; n_local elements of the basis' dtype, contiguous. */
#define LK_ACCESS_READ 0
#define LK_ACCESS_OVERWRITE 1
#define LK_ACCESS_READWRITE 2
int lk_vec_device_ptr(lk_basis_t B, int j, int access, void **dev_ptr);
/* host <-> device, `ncols` columns starting at col0; host is column-major with leading
 * dimension ldh (elements).  Synchronous. */
int lk_basis_upload(lk_basis_t B, int col0, int ncols, const void *host, int64_t ldh);
int lk_basis_download(lk_basis_t B, int col0, int ncols, void *host, int64_t ldh);

/* ---- column pool: device storage for hosts that own vectors OBJECT BY OBJECT ---------
 * LightKrylov creates vectors one object at a time, by sourced allocation, polymorphic assignment and
 * intent(out) dummies, and never frees them explicitly (AbstractVectors.fypp:595-598, gmres.fypp:110-115,155;
 * SURVEY 8b "Ownership").  A plugin type therefore cannot own device memory through allocate/final; it asks
 * this pool for a COLUMN of a shared slab (a panel of `pool_slab_cols` columns, tuning key, default 160; a single-rank
 * context takes fewer when that would exceed a quarter of the free memory, a row-sharded one never does -- the slab geometry
 * must be the same on every rank, so there an allocation that does not fit FAILS with LK_ERR_NOMEM) keyed
 * by an owner tag -- the address of the Fortran object:
 *   lk_pool_acquire  returns the column already registered to `owner_tag` (an object that reappears at the
 *                    address of a dead one re-uses its column: temporaries such as linear_combination's `proj`
 *                    cost no new memory per call), else the lowest released column, else the next column of the
 *                    open slab, so `allocate(V(k), source=b); call zero_basis(V)` lands in consecutive columns
 *                    of one panel and the lazy per-object path (lk_lazy_stats) can batch it;
 *   lk_pool_owner    tag a column is registered to (0: free or not a pool column; `slab` is validated
 *                    against the pool before it is dereferenced, so stale handles are safe to ask about);
 *   lk_pool_column_info  the same plus the column's GENERATION: a fresh value of ONE per-context counter that only grows (never
 *                    per slab, never reset by lk_pool_release_all: no value is ever handed out twice), assigned every time
 *                    lk_pool_acquire hands the column out (first use, re-use by tag, re-use after a release).  A plugin stores the generation in its
 *                    handle: a bit copy of a handle whose source object has since died and been replaced at the same address
 *                    (`allocate(b, source=dense_vector_gpu(x))` followed by another temporary) carries an old generation and
 *                    is refused instead of silently reading the new occupant's data;
 *   lk_pool_release  returns one column; lk_pool_release_all destroys every slab (after a solver call);
 *   lk_pool_stats    out4 = {slabs, columns ever carved, columns currently registered, acquisitions served by
 *                    re-use}. */
int lk_pool_acquire(lk_context_t ctx, int dtype, int64_t n_local, uint64_t owner_tag, lk_basis_t *slab,
                    int *col);
int lk_pool_owner(lk_context_t ctx, lk_basis_t slab, int col, uint64_t *owner_tag);
int lk_pool_column_info(lk_context_t ctx, lk_basis_t slab, int col, uint64_t *owner_tag, uint64_t *generation);
int lk_pool_release(lk_context_t ctx, lk_basis_t slab, int col);
int lk_pool_release_all(lk_context_t ctx);
int lk_pool_stats(lk_context_t ctx, int64_t *out4);

/* ---- abstract_vector type-bound procedures (AbstractVectors.fypp:295-381) ------------ */

/* zero(self)                         AbstractVectors.fypp:322-326, dense: 476-486 */
int lk_vec_zero(lk_basis_t B, int j);
/* rand(self, ifnorm)                 AbstractVectors.fypp:328-336, dense: 488-503.  Counter-based generator:
 * entry i (global row row0+i) = 2u-1, u = (splitmix64(seed*2^32 + ctr) >> 11) * 2^-53;
 * ctr = row for F64, 2*row / 2*row+1 for re / im.  ifnorm != 0 normalises (what eigs expects
 * of rand(.true.), IterativeSolvers.fypp:1040). */
int lk_vec_rand(lk_basis_t B, int j, uint64_t seed, int64_t row0, int ifnorm);
/* scal(self, alpha)                  AbstractVectors.fypp:338-344, dense: 505-512 */
int lk_vec_scal(lk_basis_t B, int j, const double *alpha);
/* axpby(alpha, vec, beta, self): self <- alpha*vec + beta*self   AbstractVectors.fypp:346-356,
 * dense: 514-536.
 * True axpby in ONE pass (the reference's dense version does scal-then-axpy). */
int lk_vec_axpby(const double *alpha, lk_basis_t Bx, int jx, const double *beta, lk_basis_t By,
                 int jy);
/* dot(self, vec) = sum conj(self) * vec, all-reduced over ranks.
 * AbstractVectors.fypp:358-365, dense: 538-555.
 * out: 1 or 2 doubles. */
int lk_vec_dot(lk_basis_t Bx, int jx, lk_basis_t By, int jy, double *out);
/* norm(self) = sqrt(abs(dot(self,self)))   AbstractVectors.fypp:424-432 */
int lk_vec_norm(lk_basis_t B, int j, double *out);
/* get_size(self): GLOBAL size is the caller's business; this returns the local row count.
 * AbstractVectors.fypp:367-372 */
int lk_vec_size(lk_basis_t B, int64_t *n_local);
/* copy(out, from)                    AbstractVectors.fypp:717-723 */
int lk_vec_copy(lk_basis_t Bdst, int jd, lk_basis_t Bsrc, int js);

/* ---- basis helpers built on the TBPs (cannot be specialised by a Fortran plugin; here
 *      they are fused panel kernels) ---------------------------------------------------- */

/* innerprod(X(:k), Y(jy0:jy0+p)) -> M = X^H Y, k x p column-major on the host.
 * AbstractVectors.fypp:659-695 */
int lk_innerprod(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int p, double *M);
/* linear_combination(Y, X(:k), B): Y(:, jy0+j) = X(:, :k) * C(:, j), C is k x q column-major
 * on the host.  AbstractVectors.fypp:571-643 */
int lk_lincomb(lk_basis_t Bx, int k, const double *C, int q, lk_basis_t By, int jy0);
/* Gram(X(:k)) -> G k x k (upper computed, mirrored without conjugation like the reference).
 * AbstractVectors.fypp:645-657 */
int lk_gram(lk_basis_t Bx, int k, double *G);
/* orthogonalize_against_basis(y, X(:k), info, beta=h): ONE classical Gram-Schmidt pass.
 * src/Krylov/gram_schmidt.fypp:113-154.  h: k scalars on the host (may be NULL). */
int lk_orthogonalize(lk_basis_t Bx, int k, lk_basis_t By, int jy, double *h, int *info);
/* double_gram_schmidt_step(y, X(:k), info, if_chk_orthonormal=.false., beta=h)
 * src/Krylov/gram_schmidt.fypp:12-57; interface src/Krylov/BaseKrylov.fypp:634-712.
 * Three fused panel sweeps (h1 = X^H y | y' = y - X h1, h2 = X^H y' | y'' = y' - X h2) for k <= 512 basis columns: 3k+4
 * columns of traffic (129..512 columns: the lanes of a wave are split over column groups so that a block still holds all k
 * columns of its tile); 513..2048 columns run as column panels of 512 on the device (4k - |last panel| columns, one host
 * synchronisation), wider bases panel by panel with a host round trip each.
 * h = h1 + h2 on the host (k scalars, may be NULL).
 * norms[0..2] = ||y||, ||y'||, ||y''|| (may be NULL).  info = 1 when ||y'|| < atol_dp (the
 * reference's pass-2 zero-vector flag, gram_schmidt.fypp:126-127), else 0. */
int lk_dgs(lk_basis_t Bx, int k, lk_basis_t By, int jy, double *h, double *norms, int flags,
           int *info);
/* basis-against-basis variant (gram_schmidt.fypp:59-105): Y(jy0:jy0+p) against X(:k),
 * h is k x p column-major.  Panel x panel schedules for k <= 128 -- p <= 4: the fused sweeps with several right-hand sides
 * (three passes over X per group of <= 2 columns, or of 4 columns of the real kind for k <= 64; else four);  p >= 5: the
 * coefficients and updates on the FP64 matrix cores, per group of <= 32 columns H1 = X^H Y | Y' = Y - X H1 with H2 = X^H Y' in the
 * same pass | Y'' = Y' - X H2: THREE passes over X (real kind; the complex kind keeps four: H1 | update | H2 | update).
 * Wider bases: column by column through lk_dgs. */
int lk_dgs_block(lk_basis_t Bx, int k, lk_basis_t By, int jy0, int p, double *h, int *info);

/* ---- operators (stand where a user's abstract_linop matvec/rmatvec stands,
 *      AbstractLinops.fypp:58-87); synthetic drivers of the path ------------------------ */

/* y = d .* x; d: n_local host values of the basis dtype. */
int lk_linop_diag_create(lk_context_t ctx, int dtype, int64_t n_local, const void *d_host,
                         lk_linop_t *op);
/* d_i = d0 + dstep * (row0 + i), generated on the device (F64 only). */
int lk_linop_diag_linspace_create(lk_context_t ctx, int64_t n_local, int64_t row0, double d0,
                                  double dstep, lk_linop_t *op);
/* dense_linop: y = A x ('N') or A^H x ('H'); A is n x n column-major on the host.
 * AbstractLinops.fypp:265-271, 608-660.  Whole matrix on one rank (row-sharded: the _sharded variants below). */
int lk_linop_dense_create(lk_context_t ctx, int dtype, int64_t n, const void *A_host, int64_t lda,
                          lk_linop_t *op);
/* row-sharded dense_linop ("row block of A + allgather of x"): rank r owns rows [row_starts[r], row_starts[r+1]) of
 * the n_global x n_global matrix and of every vector (row_starts: nranks + 1 entries, the same on every rank; the context's
 * rank picks the block).  A_rows: this rank's n_local x n_global block, column-major, leading dimension lda >= n_local.
 *   matvec  : x is all-gathered over the ranks (lk_allgather_fn), y_local = A_rows x -- the same products and the same
 *             summation order per row as the single-rank operator;
 *   rmatvec : z = A_rows^H x_local has n_global entries on every rank; their sum over the ranks (the all-reduce hook) is A^H x,
 *             of which each rank keeps its rows (to rounding: the single-rank operator sums a column in one piece).
 * _wrap_ takes the block where it already lies in DEVICE memory (16-byte aligned, lda even for LK_F64; not copied, not freed). */
int lk_linop_dense_create_sharded(lk_context_t ctx, int dtype, int64_t n_global, const int64_t *row_starts, const void *A_rows,
                                  int64_t lda, lk_linop_t *op);
int lk_linop_dense_wrap_sharded(lk_context_t ctx, int dtype, int64_t n_global, const int64_t *row_starts, void *dev_ptr,
                                int64_t lda, lk_linop_t *op);
/* sparse operator in CSR: y = A x ('N') or A^H x ('H'), A n x n, 0-based `rowptr[n+1]` / `colind[nnz]`, values of
 * `dtype` -- a user's sparse `abstract_linop` (AbstractLinops.fypp:58-87; the reference has no sparse type of its own:
 * its Poisson / Ginzburg-Landau examples write the stencil by hand).  The arrays are copied; A^H is built once on the
 * host so that rmatvec is a row-parallel product too.  Whole matrix on one rank (row-sharded: below). */
int lk_linop_csr_create(lk_context_t ctx, int dtype, int64_t n, const int64_t *rowptr, const int32_t *colind,
                        const void *vals, lk_linop_t *op);
/* row-sharded CSR operator: rank r owns rows [row_starts[r], row_starts[r+1]); rowptr (n_local + 1 entries, 0-based) / colind /
 * vals describe those rows with GLOBAL column indices.  COLLECTIVE: every rank calls it at the same point (the ranks exchange
 * which entries of x their rows reference, through the all-gather hook).  matvec then moves only THOSE entries -- each rank packs
 * what others need of its block, the packed pieces are all-gathered, the local rows multiply [own rows | pieces]: a stencil or
 * banded matrix exchanges a few boundary entries per neighbour, not x -- unless half of x or more would travel anyway, in which
 * case x is all-gathered whole as for the dense operator.  Either way a row's entries are summed in the same order as on one rank.
 * rmatvec sums the ranks' full-length products of their blocks' conjugate transposes (all-reduce hook) and keeps the local rows.
 * Errors are collective too: the validation of a rank's row block (column index out of range, decreasing rowptr, null arrays),
 * its buffer allocations and its uploads are each AGREED ON through the same hook before the next exchange, so when any rank fails
 * every rank returns an error and no operator (the failing rank its own message, the others "rank r failed") -- nobody is left
 * waiting in an exchange.  The one exception: a rank that cannot allocate the staging buffers of the metadata exchange itself
 * (P-length tables, the request lists) returns alone; that is fatal for the job. */
int lk_linop_csr_create_sharded(lk_context_t ctx, int dtype, int64_t n_global, const int64_t *row_starts, const int64_t *rowptr,
                                const int32_t *colind, const void *vals, lk_linop_t *op);
/* 5-point Laplacian on an N x N grid, Dirichlet, scaled by (N+1)^2 (BASELINE config 3).
 * F64; whole grid on one rank (row-sharded: the _sharded variant below). */
int lk_linop_lap5_create(lk_context_t ctx, int64_t N, lk_linop_t *op);
/* row-sharded: this rank owns grid lines [j0, j0 + nj), i.e. vector rows [j0*N, (j0+nj)*N); the neighbouring
 * ranks' boundary lines arrive through the halo exchange (one line of N doubles each way per application). */
int lk_linop_lap5_create_sharded(lk_context_t ctx, int64_t N, int64_t j0, int64_t nj, lk_linop_t *op);
/* Exponential-propagator stand-in of the Ginzburg-Landau example (BASELINE config 4): y = Phi_tau x,
 * Phi_tau = `nsub` classical RK4 steps of the linearised complex GL right-hand side with the reference's
 * stencil and boundary rows (example/ginzburg_landau/Ginzburg_Landau.f90:126-136; LK_OP_H uses the adjoint
 * right-hand side, :170-179).  x_i = -L/2 + i*dx, L = dx*(n+1); mu_i = mu_c + (mu2/2) x_i^2.
 * nu, gamma: 2 doubles each.  LK_C128; whole domain on one rank (row-sharded: the _sharded variant).  (The reference integrates with rklib's adaptive
 * rks54, an un-vendored dependency; a fixed-step RK4 is used by oracle and engine alike.) */
int lk_linop_gl_create(lk_context_t ctx, int64_t n, double dx, double tau, int nsub, const double *nu,
                       const double *gamma, double mu_c, double mu2, lk_linop_t *op);
/* row-sharded: this rank owns rows [row0, row0 + n_local) of n_global; every RK4 stage exchanges one point with each
 * neighbouring rank through the halo exchange. */
int lk_linop_gl_create_sharded(lk_context_t ctx, int64_t n_global, int64_t row0, int64_t n_local, double dx,
                               double tau, int nsub, const double *nu, const double *gamma, double mu_c,
                               double mu2, lk_linop_t *op);
int lk_linop_destroy(lk_linop_t op);
/* apply_matvec / apply_rmatvec: y(:, jy) = op(A) x(:, jx).  AbstractLinops.fypp:391-424 */
int lk_linop_apply(lk_linop_t op, int trans, lk_basis_t Bx, int jx, lk_basis_t By, int jy);

/* ---- the caller of the path: Arnoldi --------------------------------------------------
 * arnoldi(A, X, H, info, kstart, kend, tol, transpose) with blksize = 1 (lk_arnoldi_block: any blksize).
 * src/Krylov/arnoldi.fypp:8-76 (+ the 1-column qr_no_pivoting, src/Krylov/qr.fypp:116-167).
 * X: basis with m+1 columns; H: host (ldh x m) column-major array of the basis dtype;
 * kstart/kend 1-based inclusive; tol: breakdown tolerance (reference default atol_dp).
 * info = 0, or k when |H(k+1,k)| < tol (invariant subspace, loop exits).  Steps up to 512 basis columns are enqueued
 * asynchronously (one host synchronisation per call); beyond that one round trip per step. */
int lk_arnoldi(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int kstart, int kend,
               double tol, int trans, int *info);
/* The same factorisation (same bits), delivered IN SEGMENTS while it runs: the steps are enqueued back to back as one asynchronous batch,
 * and as soon as the device has produced the steps up to seg_last[i] (nseg ascending step numbers within [kstart, kend]) their columns of H
 * are written and fn(user, kfirst, klast) is called on the calling thread -- while the later steps are still running on the device.  The
 * per-step host work of a caller overlaps the rest of the call with no idle gap on the device between segments: eigs tests the Ritz pairs of
 * H(1:k, 1:k) after every step (src/IterativeSolvers/IterativeSolvers.fypp:1059-1093); one blocking lk_arnoldi call per segment cost its
 * Krylov-Schur cycle ~0.4 ms of device idling per boundary.  fn is called for every step range exactly once, in order, the last time
 * before the call returns; columns beyond a breakdown are never reported.  fn returns 0 to go on, non-zero to STOP: nothing more is
 * enqueued (the device is kept at most 24 steps ahead of the segment being delivered, so at most that many steps beyond the last
 * reported one have touched the basis), nothing more is reported, and the call returns with info = 0 (eigs stops a cycle at the first
 * step with enough converged pairs, :1087-1093).  fn must not call into the same context.  fn may be NULL (then this is lk_arnoldi).
 * ROW-SHARDED CONTEXTS (nranks > 1): every step carries three all-reduces, and each rank stops enqueueing where ITS fn says so -- the
 * library has no agreement step here (one more collective per delivery would drain the 24 steps of lookahead this entry exists for).
 * So the value fn returns for a given (kfirst, klast) must be THE SAME ON EVERY RANK: a function of H (identical on all ranks after the
 * all-reduce) is; wall-clock time, or state set asynchronously by another thread, is not -- ranks that stop at different steps leave
 * unmatched collectives behind (a hang, or sums paired across steps).  The mirror's eigs never asks a sharded cycle to stop early. */
typedef int (*lk_progress_fn)(void *user, int kfirst, int klast);
int lk_arnoldi_segments(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int kstart, int kend, double tol, int trans,
                        const int *seg_last, int nseg, lk_progress_fn fn, void *user, int *info);

/* ---- Lanczos tridiagonalisation (symmetric / Hermitian operators) ------------------------
 * lanczos_tridiagonalization(A, X, T, info, kstart, kend, tol): src/Krylov/lanczos.fypp:7-64.
 * Per step: X(k+1) = A X(k); T(i, k) = X(i)%dot(X(k+1)), X(k+1) -= T(i, k) X(i) for i = max(1, k-1), k (:57-60); full
 * re-orthogonalisation by a double Gram-Schmidt step without beta (:62); T(k+1, k) = ||X(k+1)||; beta < tol => info = k and
 * the loop exits WITHOUT scaling (:32-36), else X(k+1) is normalised.  All steps of a call are enqueued asynchronously
 * (device-side stop flag), one host synchronisation per call.  T: host (ldt x m) column-major array of the basis dtype;
 * only T(k-1:k+1, k) of each step is written.  Steps against more than 512 basis columns run one host round trip each (the reference has no
 * cap, lanczos.fypp:20). */
int lk_lanczos(lk_linop_t A, lk_basis_t X, double *T, int64_t ldt, int kstart, int kend, double tol, int *info);

/* ---- Golub-Kahan bidiagonalisation -----------------------------------------------------------
 * lanczos_bidiagonalization(A, U, V, B, info, kstart, kend, tol): src/Krylov/golub_kahan.fypp:7-64.
 * Per step: V(k) = A^H U(k), double Gram-Schmidt against V(:k-1) (k > 1), B(k, k) = alpha = ||V(k)||, normalise (:27-42);
 * U(k+1) = A V(k), double Gram-Schmidt against U(:k), B(k+1, k) = beta = ||U(k+1)||, normalise (:45-58); a norm not above
 * tol => info = k and the loop exits without scaling.  All steps of a call are enqueued asynchronously (device-side stop flag
 * per half step), one host synchronisation per call.  U: kdim + 1 columns, V: >= kdim columns (two different bases); B: host
 * (ldb x kdim) column-major array of the basis dtype, only B(k, k) and B(k+1, k) are written.  tol >= atol_dp.  Steps against more than
 * 512 basis columns run one host round trip each (the reference has no cap, golub_kahan.fypp:18). */
int lk_bidiag(lk_linop_t A, lk_basis_t U, lk_basis_t V, double *B, int64_t ldb, int kstart, int kend, double tol, int *info);

/* ---- qr_no_pivoting and the block Arnoldi factorisation (round 6) ------------------------------------------
 * qr_no_pivoting(Q, R, info, tol): src/Krylov/qr.fypp:116-167, on columns [j0, j0 + p) of a panel.  Column j: double Gram-Schmidt
 * against the j columns before it with beta = R(:j-1, j) (:131-134), beta = ||q_j|| (:135), NaN aborts (:137-143); beta < tol => info = j
 * (first such column), R(j, j) = 0, the column is re-drawn from the counter generator (stream 0x5EED + panel column + 1), orthogonalised
 * again and its new norm taken (:146-162), else R(j, j) = beta; q_j scaled by 1 / beta (:164).  R: host p x p column-major (leading
 * dimension ldr elements) of the basis dtype, zeroed by the call.  Host-synchronous, column by column. */
int lk_qr(lk_basis_t Q, int j0, int p, double *R, int64_t ldr, double tol, int *info);
/* arnoldi(A, X, H, info, kstart, kend, tol, transpose, blksize): src/Krylov/arnoldi.fypp:8-76 with blksize = p > 1 (p = 1 is lk_arnoldi).
 * X holds (kdim + 1) p columns (kdim = (ncols - p) / p, :26), H is host ((kdim + 1) p x kdim p), column-major, leading dimension ldh.
 * Step k (kpm = (k-1) p, kp = k p): X(kp + i) = A X(kpm + i) (:39-47), the batch Gram-Schmidt of the new block against X(:kp) into
 * H(:kp, kpm+1:kp) (:50-51, panel x panel on the matrix cores from five columns on), qr_no_pivoting of the block into
 * H(kp+1:kp+p, kpm+1:kp) (:55), min |diagonal| < tol => info = kp, exit (:58-71).  All steps of a call are enqueued asynchronously
 * behind a device-side stop flag that counts (step, column): a column below max(tol, atol_dp) stops everything behind it, the host
 * finishes that block (re-draw, remaining columns) exactly as lk_qr would; ONE copy and ONE host synchronisation per call otherwise.
 * Bases beyond 512 columns, and "async_arnoldi" = 0, run one round trip per step through lk_dgs_block / lk_qr. */
int lk_arnoldi_block(lk_linop_t A, lk_basis_t X, double *H, int64_t ldh, int blksize, int kstart, int kend, double tol, int trans,
                     int *info);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTKRYLOV_HIP_H */
