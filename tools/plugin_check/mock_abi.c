/* mock_abi.c -- HOST stand-in for the subset of include/lightkrylov_hip.h that fortran/dense_vector_gpu.f90 calls.
 *
 * BUILD-CONTAINER TOOLING for tools/check_plugin.sh ONLY.  It is not part of the product, is never loaded by
 * lightkrylov_amd, does not travel to the GPU box, and is not an oracle (nothing is compared with it for parity).
 * Its one job: let the Fortran plugin's OBJECT-SEMANTICS logic (owner tags, copy-on-write of bit copies, deep-copy
 * assignment, intent(out) re-acquisition, column re-use) execute under the reference's own arnoldi / gmres control
 * flow in a container that has no GPU, so that bugs of that logic show up as wrong numbers or a growing pool here
 * instead of on a user's machine.  Vectors are plain host arrays; arithmetic is the obvious loop.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LK_F64 0
#define LK_C128 1
#define LK_OK 0
#define LK_ERR_INVALID (-1)

typedef struct mock_basis { int dtype; int64_t n; int ncols; double *data; } mock_basis;
typedef struct mock_op { int kind; int dtype; int64_t n; double *a; } mock_op;   /* kind 0 diag, 1 dense */
#define MAX_SLABS 64
static int g_in_pool = 0;   /* slabs are created / destroyed by the pool itself: not ABI calls of the plugin */
typedef struct mock_ctx {
    mock_basis *slab[MAX_SLABS]; int used[MAX_SLABS]; uint64_t *owner[MAX_SLABS]; uint64_t *gen[MAX_SLABS]; int nslabs;
    int slab_cols; int64_t carved, reused; int64_t row0;
} mock_ctx;

/* ---- optional call trace (LK_MOCK_TRACE=<file>): every ABI call the plugin makes under the reference's solvers, with the
 * values the mock returned (dot products, downloads, pool placements).  tests/golden/plugin_abi_trace.txt.gz is such a trace;
 * tests/test_gpu_plugin_trace.py replays it against the real engine on the GPU.  A trace is DATA (a call sequence and numbers). */
static FILE *g_tr = NULL;
static int g_tr_init = 0;
#define MAX_IDS 4096
static const void *g_ids[MAX_IDS];
static int g_nids = 0;
static FILE *tr(void) {
    if (!g_tr_init) { g_tr_init = 1; const char *f = getenv("LK_MOCK_TRACE"); if (f && *f) g_tr = fopen(f, "w"); }
    return g_tr;
}
static int idof(const void *p) {          /* stable small integer per object (basis, slab, operator) */
    for (int i = 0; i < g_nids; ++i) if (g_ids[i] == p) return i;
    if (g_nids < MAX_IDS) { g_ids[g_nids] = p; return g_nids++; }
    return -1;
}
static void forget(const void *p) { for (int i = 0; i < g_nids; ++i) if (g_ids[i] == p) g_ids[i] = (const void *)(intptr_t)-1 - i; }
static void trv(const double *v, int64_t cnt) { for (int64_t i = 0; i < cnt; ++i) fprintf(g_tr, " %.17g", v[i]); }
/* LK_MOCK_TRACE_PAUSE=1 (set and cleared by the driver around a check) keeps a stretch of calls out of the trace */
static int paused(void) { const char *p = getenv("LK_MOCK_TRACE_PAUSE"); return p && *p == '1'; }
#define TR(...) do { if (tr() && !paused()) { fprintf(g_tr, __VA_ARGS__); } } while (0)
#define TRNL() do { if (g_tr) { fputc('\n', g_tr); } } while (0)

static char g_err[256] = "";
static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "mock: %s", msg); return LK_ERR_INVALID; }
static int ed(const mock_basis *b) { return b->dtype == LK_C128 ? 2 : 1; }
static double *col(const mock_basis *b, int j) { return b->data + (int64_t)j * b->n * ed(b); }

const char *lk_last_error(void) { return g_err; }
int lk_version(void) { return 100; }
int lk_init(int device, void *stream, mock_ctx **ctx) { (void)device; (void)stream; *ctx = calloc(1, sizeof(mock_ctx)); (*ctx)->slab_cols = 160; TR("init\n"); return LK_OK; }
int lk_set_tuning(mock_ctx *c, const char *key, int v) { if (!strcmp(key, "pool_slab_cols")) c->slab_cols = v; TR("tuning %s %d\n", key, v); return LK_OK; }
int lk_set_partition(mock_ctx *c, int64_t row0, int64_t ng) { (void)ng; c->row0 = row0; return LK_OK; }
int lk_sync(mock_ctx *c) { (void)c; return LK_OK; }

int lk_basis_create(mock_ctx *c, int dtype, int64_t n, int ncols, mock_basis **B) {
    (void)c;
    mock_basis *b = malloc(sizeof *b);
    b->dtype = dtype; b->n = n; b->ncols = ncols;
    b->data = calloc((size_t)(n > 0 ? n : 1) * ncols * (dtype == LK_C128 ? 2 : 1), sizeof(double));
    *B = b;
    if (!g_in_pool) TR("basis_create %d %lld %d -> %d\n", dtype, (long long)n, ncols, idof(b));
    return LK_OK;
}
int lk_basis_destroy(mock_basis *b) { if (b) { if (!g_in_pool) TR("basis_destroy %d\n", idof(b)); forget(b); free(b->data); free(b); } return LK_OK; }
int lk_basis_upload(mock_basis *b, int c0, int nc, const void *host, int64_t ldh) {
    for (int j = 0; j < nc; ++j) memcpy(col(b, c0 + j), (const double *)host + (int64_t)j * ldh * ed(b), (size_t)b->n * ed(b) * 8);
    if (tr() && !paused()) { fprintf(g_tr, "upload %d %d %d", idof(b), c0, nc); for (int j = 0; j < nc; ++j) trv(col(b, c0 + j), b->n * ed(b)); TRNL(); }
    return LK_OK;
}
int lk_basis_download(mock_basis *b, int c0, int nc, void *host, int64_t ldh) {
    for (int j = 0; j < nc; ++j) memcpy((double *)host + (int64_t)j * ldh * ed(b), col(b, c0 + j), (size_t)b->n * ed(b) * 8);
    if (tr() && !paused()) { fprintf(g_tr, "download %d %d %d ->", idof(b), c0, nc); for (int j = 0; j < nc; ++j) trv(col(b, c0 + j), b->n * ed(b)); TRNL(); }
    return LK_OK;
}

/* ---- pool: same contract as the engine's (lowest free column first, re-use by owner tag) ---- */
static int find_slab(mock_ctx *c, mock_basis *s) { for (int i = 0; i < c->nslabs; ++i) if (c->slab[i] == s) return i; return -1; }
int lk_pool_acquire(mock_ctx *c, int dtype, int64_t n, uint64_t tag, mock_basis **slab, int *colo) {
    if (!tag) return fail("tag 0");
    for (int i = 0; i < c->nslabs; ++i)
        for (int j = 0; j < c->used[i]; ++j)
            if (c->owner[i][j] == tag) {
                if (c->slab[i]->dtype == dtype && c->slab[i]->n == n) { *slab = c->slab[i]; *colo = j; c->reused++; c->gen[i][j]++; TR("pool_acquire %d %lld %llu -> %d %d\n", dtype, (long long)n, (unsigned long long)tag, idof(*slab), j); return LK_OK; }
                c->owner[i][j] = 0;
            }
    for (int i = 0; i < c->nslabs; ++i)
        if (c->slab[i]->dtype == dtype && c->slab[i]->n == n)
            for (int j = 0; j < c->used[i]; ++j)
                if (c->owner[i][j] == 0) { c->owner[i][j] = tag; c->gen[i][j]++; *slab = c->slab[i]; *colo = j; c->reused++; TR("pool_acquire %d %lld %llu -> %d %d\n", dtype, (long long)n, (unsigned long long)tag, idof(*slab), j); return LK_OK; }
    int si = -1;
    for (int i = c->nslabs - 1; i >= 0; --i)
        if (c->slab[i]->dtype == dtype && c->slab[i]->n == n && c->used[i] < c->slab[i]->ncols) { si = i; break; }
    if (si < 0) {
        if (c->nslabs == MAX_SLABS) return fail("too many slabs");
        si = c->nslabs++;
        g_in_pool = 1; lk_basis_create(c, dtype, n, c->slab_cols, &c->slab[si]); g_in_pool = 0;
        c->owner[si] = calloc((size_t)c->slab_cols, sizeof(uint64_t));
        c->gen[si] = calloc((size_t)c->slab_cols, sizeof(uint64_t));
        c->used[si] = 0;
    }
    const int j = c->used[si]++;
    c->owner[si][j] = tag; c->gen[si][j]++; c->carved++;
    *slab = c->slab[si]; *colo = j;
    TR("pool_acquire %d %lld %llu -> %d %d\n", dtype, (long long)n, (unsigned long long)tag, idof(*slab), j);
    return LK_OK;
}
int lk_pool_owner(mock_ctx *c, mock_basis *slab, int j, uint64_t *tag) {
    *tag = 0;
    const int si = find_slab(c, slab);
    if (si >= 0 && j >= 0 && j < c->used[si]) *tag = c->owner[si][j];
    return LK_OK;
}
int lk_pool_column_info(mock_ctx *c, mock_basis *slab, int j, uint64_t *tag, uint64_t *gen) {
    if (tag) *tag = 0;
    if (gen) *gen = 0;
    const int si = find_slab(c, slab);
    if (si >= 0 && j >= 0 && j < c->used[si]) { if (tag) *tag = c->owner[si][j]; if (gen) *gen = c->gen[si][j]; }
    return LK_OK;
}
int lk_pool_release(mock_ctx *c, mock_basis *slab, int j) {
    const int si = find_slab(c, slab);
    if (si < 0) return fail("not a pool column");
    c->owner[si][j] = 0; TR("pool_release %d %d\n", idof(slab), j); return LK_OK;
}
int lk_pool_release_all(mock_ctx *c) {
    TR("pool_release_all\n");
    g_in_pool = 1;
    for (int i = 0; i < c->nslabs; ++i) { lk_basis_destroy(c->slab[i]); free(c->owner[i]); free(c->gen[i]); }
    g_in_pool = 0;
    c->nslabs = 0; return LK_OK;
}
int lk_pool_stats(mock_ctx *c, int64_t *o) {
    int64_t live = 0;
    for (int i = 0; i < c->nslabs; ++i) for (int j = 0; j < c->used[i]; ++j) live += c->owner[i][j] != 0;
    o[0] = c->nslabs; o[1] = c->carved; o[2] = live; o[3] = c->reused; return LK_OK;
}
int lk_finalize(mock_ctx *c) { if (c) { lk_pool_release_all(c); free(c); } TR("finalize\n"); if (g_tr) fflush(g_tr); return LK_OK; }

/* ---- vector primitives ---- */
int lk_vec_zero(mock_basis *b, int j) { memset(col(b, j), 0, (size_t)b->n * ed(b) * 8); TR("zero %d %d\n", idof(b), j); return LK_OK; }
static int g_in_rand = 0;
static uint64_t splitmix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int lk_vec_dot(mock_basis *bx, int jx, mock_basis *by, int jy, double *out) {
    if (bx->dtype == LK_C128) {
        const double _Complex *x = (const double _Complex *)col(bx, jx), *y = (const double _Complex *)col(by, jy);
        double _Complex s = 0; for (int64_t i = 0; i < bx->n; ++i) s += conj(x[i]) * y[i];
        out[0] = creal(s); out[1] = cimag(s);
    } else {
        const double *x = col(bx, jx), *y = col(by, jy);
        double s = 0; for (int64_t i = 0; i < bx->n; ++i) s += x[i] * y[i];
        out[0] = s;
    }
    if (!g_in_rand) TR("dot %d %d %d %d -> %.17g %.17g\n", idof(bx), jx, idof(by), jy, out[0], bx->dtype == LK_C128 ? out[1] : 0.0);
    return LK_OK;
}
int lk_vec_scal(mock_basis *b, int j, const double *a) {
    if (b->dtype == LK_C128) { double _Complex *x = (double _Complex *)col(b, j), al = a[0] + a[1] * I; for (int64_t i = 0; i < b->n; ++i) x[i] *= al; }
    else { double *x = col(b, j); for (int64_t i = 0; i < b->n; ++i) x[i] *= a[0]; }
    if (!g_in_rand) TR("scal %d %d %.17g %.17g\n", idof(b), j, a[0], b->dtype == LK_C128 ? a[1] : 0.0);
    return LK_OK;
}
int lk_vec_rand(mock_basis *b, int j, uint64_t seed, int64_t row0, int ifnorm) {
    double *x = col(b, j);
    for (int64_t i = 0; i < b->n * ed(b); ++i) x[i] = 2.0 * ((double)(splitmix64((seed << 32) + (uint64_t)(row0 * ed(b) + i)) >> 11) * 0x1.0p-53) - 1.0;
    g_in_rand = 1;
    if (ifnorm) { double d[2]; lk_vec_dot(b, j, b, j, d); double s[2] = {1.0 / sqrt(d[0]), 0.0}; lk_vec_scal(b, j, s); }
    g_in_rand = 0;
    TR("rand %d %d %llu %lld %d\n", idof(b), j, (unsigned long long)seed, (long long)row0, ifnorm);
    return LK_OK;
}
int lk_vec_axpby(const double *a, mock_basis *bx, int jx, const double *bt, mock_basis *by, int jy) {
    if (bx->n != by->n) return fail("size mismatch");
    if (bx->dtype == LK_C128) {
        const double _Complex *x = (const double _Complex *)col(bx, jx); double _Complex *y = (double _Complex *)col(by, jy);
        const double _Complex al = a[0] + a[1] * I, be = bt[0] + bt[1] * I;
        const int bz = (bt[0] == 0.0 && bt[1] == 0.0);
        for (int64_t i = 0; i < bx->n; ++i) y[i] = bz ? al * x[i] : al * x[i] + be * y[i];
    } else {
        const double *x = col(bx, jx); double *y = col(by, jy);
        for (int64_t i = 0; i < bx->n; ++i) y[i] = (bt[0] == 0.0) ? a[0] * x[i] : a[0] * x[i] + bt[0] * y[i];
    }
    { const int cz = bx->dtype == LK_C128; TR("axpby %.17g %.17g %d %d %.17g %.17g %d %d\n", a[0], cz ? a[1] : 0.0, idof(bx), jx, bt[0], cz ? bt[1] : 0.0, idof(by), jy); }
    return LK_OK;
}
int lk_vec_copy(mock_basis *bd, int jd, mock_basis *bs, int js) { memmove(col(bd, jd), col(bs, js), (size_t)bd->n * ed(bd) * 8); TR("copy %d %d %d %d\n", idof(bd), jd, idof(bs), js); return LK_OK; }

/* ---- operators: diagonal and dense only ---- */
int lk_linop_diag_create(mock_ctx *c, int dtype, int64_t n, const void *d, mock_op **op) {
    (void)c; mock_op *o = malloc(sizeof *o); o->kind = 0; o->dtype = dtype; o->n = n;
    const size_t bytes = (size_t)n * (dtype == LK_C128 ? 16 : 8); o->a = malloc(bytes); memcpy(o->a, d, bytes); *op = o;
    if (tr() && !paused()) { fprintf(g_tr, "op_diag %d %lld -> %d :", dtype, (long long)n, idof(o)); trv(o->a, (int64_t)(bytes / 8)); TRNL(); }
    return LK_OK;
}
int lk_linop_dense_create(mock_ctx *c, int dtype, int64_t n, const void *A, int64_t lda, mock_op **op) {
    (void)c; if (lda != n) return fail("lda != n");
    mock_op *o = malloc(sizeof *o); o->kind = 1; o->dtype = dtype; o->n = n;
    const size_t bytes = (size_t)n * n * (dtype == LK_C128 ? 16 : 8); o->a = malloc(bytes); memcpy(o->a, A, bytes); *op = o;
    if (tr() && !paused()) { fprintf(g_tr, "op_dense %d %lld -> %d :", dtype, (long long)n, idof(o)); trv(o->a, (int64_t)(bytes / 8)); TRNL(); }
    return LK_OK;
}
int lk_vec_device_ptr(mock_basis *b, int j, int access, void **p) { (void)access; *p = col(b, j); return LK_OK; }
int lk_linop_dense_create_sharded(mock_ctx *c, int dtype, int64_t n, const int64_t *rs, const void *A, int64_t lda, mock_op **op) {
    if (rs[0] != 0 || rs[1] != n) return fail("the mock is one rank");
    return lk_linop_dense_create(c, dtype, n, A, lda, op);
}
int lk_linop_csr_create_sharded(mock_ctx *c, int dtype, int64_t n, const int64_t *rs, const int64_t *rp, const int32_t *ci, const void *v, mock_op **op) { (void)c; (void)dtype; (void)n; (void)rs; (void)rp; (void)ci; (void)v; (void)op; return fail("not in the mock"); }
int lk_linop_csr_create(mock_ctx *c, int dtype, int64_t n, const int64_t *rp, const int32_t *ci, const void *v, mock_op **op) { (void)c; (void)dtype; (void)n; (void)rp; (void)ci; (void)v; (void)op; return fail("not in the mock"); }
int lk_linop_diag_linspace_create(mock_ctx *c, int64_t n, int64_t r0, double d0, double ds, mock_op **op) { (void)c; (void)n; (void)r0; (void)d0; (void)ds; (void)op; return fail("not in the mock"); }
int lk_linop_lap5_create(mock_ctx *c, int64_t N, mock_op **op) { (void)c; (void)N; (void)op; return fail("not in the mock"); }
int lk_linop_gl_create(mock_ctx *c, int64_t n, double dx, double tau, int nsub, const double *nu, const double *ga, double mc, double m2, mock_op **op) {
    (void)c; (void)n; (void)dx; (void)tau; (void)nsub; (void)nu; (void)ga; (void)mc; (void)m2; (void)op; return fail("not in the mock");
}
int lk_linop_lap5_create_sharded(mock_ctx *c, int64_t N, int64_t j0, int64_t nj, mock_op **op) { (void)c; (void)N; (void)j0; (void)nj; (void)op; return fail("not in the mock"); }
int lk_linop_gl_create_sharded(mock_ctx *c, int64_t ng, int64_t r0, int64_t nl, double dx, double tau, int nsub, const double *nu, const double *ga, double mc, double m2, mock_op **op) {
    (void)c; (void)ng; (void)r0; (void)nl; (void)dx; (void)tau; (void)nsub; (void)nu; (void)ga; (void)mc; (void)m2; (void)op; return fail("not in the mock");
}
int lk_linop_destroy(mock_op *o) { if (o) { TR("op_destroy %d\n", idof(o)); forget(o); free(o->a); free(o); } return LK_OK; }
int lk_linop_apply(mock_op *o, int trans, mock_basis *bx, int jx, mock_basis *by, int jy) {
    const int64_t n = o->n;
    if (col(bx, jx) == col(by, jy)) return fail("vec_in and vec_out alias");
    if (o->dtype == LK_C128) {
        const double _Complex *a = (const double _Complex *)o->a, *x = (const double _Complex *)col(bx, jx); double _Complex *y = (double _Complex *)col(by, jy);
        for (int64_t i = 0; i < n; ++i) {
            if (o->kind == 0) { y[i] = (trans ? conj(a[i]) : a[i]) * x[i]; continue; }
            double _Complex s = 0;
            for (int64_t j = 0; j < n; ++j) s += (trans ? conj(a[i * n + j]) : a[j * n + i]) * x[j];
            y[i] = s;
        }
    } else {
        const double *a = o->a, *x = col(bx, jx); double *y = col(by, jy);
        for (int64_t i = 0; i < n; ++i) {
            if (o->kind == 0) { y[i] = a[i] * x[i]; continue; }
            double s = 0;
            for (int64_t j = 0; j < n; ++j) s += (trans ? a[i * n + j] : a[j * n + i]) * x[j];
            y[i] = s;
        }
    }
    TR("op_apply %d %d %d %d %d %d\n", idof(o), trans, idof(bx), jx, idof(by), jy);
    return LK_OK;
}
int lk_arnoldi(void *A, void *X, double *H, int64_t ldh, int k0, int k1, double tol, int trans, int *info) {
    (void)A; (void)X; (void)H; (void)ldh; (void)k0; (void)k1; (void)tol; (void)trans; (void)info; return fail("lk_arnoldi is not in the mock");
}
int lk_arnoldi_block(void *A, void *X, double *H, int64_t ldh, int p, int k0, int k1, double tol, int trans, int *info) {
    (void)A; (void)X; (void)H; (void)ldh; (void)p; (void)k0; (void)k1; (void)tol; (void)trans; (void)info; return fail("lk_arnoldi_block is not in the mock");
}
int lk_arnoldi_segments(void *A, void *X, double *H, int64_t ldh, int k0, int k1, double tol, int trans, const int *segs, int nseg, void *fn, void *user, int *info) {
    (void)A; (void)X; (void)H; (void)ldh; (void)k0; (void)k1; (void)tol; (void)trans; (void)segs; (void)nseg; (void)fn; (void)user; (void)info;
    return fail("lk_arnoldi_segments is not in the mock");
}
int lk_lanczos(void *A, void *X, double *T, int64_t ldt, int k0, int k1, double tol, int *info) {
    (void)A; (void)X; (void)T; (void)ldt; (void)k0; (void)k1; (void)tol; (void)info; return fail("lk_lanczos is not in the mock");
}
int lk_bidiag(void *A, void *U, void *V, double *B, int64_t ldb, int k0, int k1, double tol, int *info) {
    (void)A; (void)U; (void)V; (void)B; (void)ldb; (void)k0; (void)k1; (void)tol; (void)info; return fail("lk_bidiag is not in the mock");
}
int lk_qr(void *Q, int j0, int p, double *R, int64_t ldr, double tol, int *info) {
    (void)Q; (void)j0; (void)p; (void)R; (void)ldr; (void)tol; (void)info; return fail("lk_qr is not in the mock");
}
int lk_comm_get_unique_id(void *id) { memset(id, 0, 128); return LK_OK; }
int lk_comm_init_rank(mock_ctx *c, int nranks, int rank, const void *id) { (void)c; (void)rank; (void)id; return nranks == 1 ? LK_OK : fail("no collective in the mock"); }
int lk_comm_info(mock_ctx *c, int *nranks, int *rank) { (void)c; if (nranks) *nranks = 1; if (rank) *rank = 0; return LK_OK; }
