/* Plain-C client of the C ABI (no Python, no torch, no Fortran): proves the boundary is usable as a
 * C library.  Runs a 16-step Arnoldi on a diagonal operator, checks orthonormality through lk_gram,
 * the Arnoldi relation on the host, error paths, and wrap of caller-owned device memory.
 * Built and run by tests/test_c_client.py (gcc + libamdhip64 for hipMalloc only). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/lightkrylov_hip.h"

/* minimal HIP prototypes so this file compiles with gcc (no hip headers needed) */
extern int hipMalloc(void **p, size_t n);
extern int hipFree(void *p);
extern int hipMemset(void *p, int v, size_t n);
extern int hipDeviceSynchronize(void);

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != LK_OK) { printf("FAIL %s -> %d: %s\n", #call, rc_, lk_last_error()); return 1; } \
    } while (0)
#define EXPECT(cond)                                                                   \
    do { if (!(cond)) { printf("FAIL expectation %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

/* progress function of lk_arnoldi_segments: records the ranges reported; asks to stop once column `stop_at` (> 0) has been reported */
typedef struct { int calls, first[32], last[32], stop_at; } progress_log;
static int on_columns(void *user, int kfirst, int klast) {
    progress_log *p = (progress_log *)user;
    if (p->calls < 32) { p->first[p->calls] = kfirst; p->last[p->calls] = klast; }
    p->calls += 1;
    return p->stop_at > 0 && klast >= p->stop_at;
}

int main(void) {
    const int64_t n = 50001; const int m = 16;
    lk_context_t ctx; lk_basis_t X, W; lk_linop_t A;
    CHECK(lk_init(0, NULL, &ctx));
    EXPECT(lk_version() >= 100);

    double *d = malloc(n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) d[i] = 1.0 + (double)i / (double)n;
    CHECK(lk_basis_create(ctx, LK_F64, n, m + 1, &X));
    CHECK(lk_vec_rand(X, 0, 7, 0, 1));
    CHECK(lk_linop_diag_create(ctx, LK_F64, n, d, &A));
    double *H = calloc((size_t)(m + 1) * m, sizeof(double));
    int info = -99;
    CHECK(lk_arnoldi(A, X, H, m + 1, 1, m, 1e-15, 0, &info));
    EXPECT(info == 0);

    /* the same factorisation delivered in segments (round 5): same H bit for bit, every step reported once, in order; stop on request */
    {
        lk_basis_t X2; int info2 = -99, nranks = -1, rank = -1;
        CHECK(lk_comm_info(ctx, &nranks, &rank)); EXPECT(nranks == 1 && rank == 0);
        CHECK(lk_basis_create(ctx, LK_F64, n, m + 1, &X2));
        CHECK(lk_vec_rand(X2, 0, 7, 0, 1));
        double *H2 = calloc((size_t)(m + 1) * m, sizeof(double));
        const int segs[3] = {4, 10, 15};
        progress_log log = {0};
        CHECK(lk_arnoldi_segments(A, X2, H2, m + 1, 1, m, 1e-15, 0, segs, 3, on_columns, &log, &info2));
        EXPECT(info2 == 0 && memcmp(H, H2, (size_t)(m + 1) * m * sizeof(double)) == 0);
        EXPECT(log.calls == 4 && log.first[0] == 1 && log.last[0] == 4 && log.first[1] == 5 && log.last[1] == 10 && log.first[2] == 11 &&
               log.last[2] == 15 && log.first[3] == 16 && log.last[3] == m);
        memset(H2, 0, (size_t)(m + 1) * m * sizeof(double));
        CHECK(lk_vec_rand(X2, 0, 7, 0, 1));
        progress_log log2 = {0}; log2.stop_at = 4;
        CHECK(lk_arnoldi_segments(A, X2, H2, m + 1, 1, m, 1e-15, 0, segs, 3, on_columns, &log2, &info2));
        EXPECT(info2 == 0 && log2.calls == 1 && memcmp(H, H2, (size_t)(m + 1) * 4 * sizeof(double)) == 0);   /* columns 1..4, nothing reported after the request */
        EXPECT(lk_arnoldi_segments(A, X2, H2, m + 1, 1, m, 1e-15, 0, (const int[]){10, 4}, 2, on_columns, &log2, &info2) == LK_ERR_INVALID);   /* not ascending */
        free(H2); lk_basis_destroy(X2);
    }

    /* orthonormality via lk_gram */
    double *G = malloc((size_t)(m + 1) * (m + 1) * sizeof(double));
    CHECK(lk_gram(X, m + 1, G));
    double orth = 0;
    for (int i = 0; i <= m; ++i) for (int j = 0; j <= m; ++j) {
        double e = fabs(G[i + j * (m + 1)] - (i == j ? 1.0 : 0.0)); if (e > orth) orth = e; }
    EXPECT(orth < 1e-13);

    /* A X_m = X_{m+1} H on the host */
    double *Xh = malloc((size_t)n * (m + 1) * sizeof(double));
    CHECK(lk_basis_download(X, 0, m + 1, Xh, n));
    double res = 0;
    for (int j = 0; j < m; ++j) for (int64_t i = 0; i < n; i += 97) {
        double s = d[i] * Xh[i + (size_t)j * n];
        for (int l = 0; l <= m; ++l) s -= Xh[i + (size_t)l * n] * H[l + j * (m + 1)];
        if (fabs(s) > res) res = fabs(s); }
    EXPECT(res < 1e-13);

    /* dot / norm / axpby through (basis, column) pairs */
    double dd[2], nn;
    CHECK(lk_vec_dot(X, 0, X, 1, dd)); EXPECT(fabs(dd[0]) < 1e-13);
    CHECK(lk_vec_norm(X, 3, &nn)); EXPECT(fabs(nn - 1.0) < 1e-14);

    /* caller-owned device memory: wrap, ld == n rounded to even, garbage beyond n_local is never read */
    const int64_t ld = n + 1; void *dev = NULL;
    EXPECT(hipMalloc(&dev, (size_t)ld * 2 * sizeof(double)) == 0);
    EXPECT(hipMemset(dev, 0xFF, (size_t)ld * 2 * sizeof(double)) == 0);   /* NaN pattern everywhere */
    EXPECT(hipDeviceSynchronize() == 0);   /* the memset runs on the null stream, the engine on its own */
    CHECK(lk_basis_wrap(ctx, LK_F64, n, 2, ld, dev, &W));
    CHECK(lk_vec_copy(W, 0, X, 0));
    CHECK(lk_vec_copy(W, 1, X, 1));
    double alpha = 2.0, beta = -1.0;
    CHECK(lk_vec_axpby(&alpha, W, 0, &beta, W, 1));          /* W1 = 2 X0 - X1 */
    CHECK(lk_vec_norm(W, 1, &nn)); EXPECT(fabs(nn - sqrt(5.0)) < 1e-13);
    double h[1], norms[3]; int dinfo;
    CHECK(lk_dgs(W, 1, W, 1, h, norms, 0, &dinfo));           /* orthogonalise W1 against W0 */
    EXPECT(fabs(h[0] - 2.0) < 1e-13 && fabs(norms[2] - 1.0) < 1e-13 && dinfo == 0);

    /* round 6: qr_no_pivoting and the block Arnoldi factorisation as engine calls (arnoldi.fypp:20-73 with blksize = 2; qr.fypp:116-167):
     * A X = X+ H+ on every column, X orthonormal, the single-launch step's statistics available, continued ranges = the one-shot run */
    {
        const int p = 2, kd = 6, nc = (kd + 1) * p;
        lk_basis_t XB; int binfo = -99, qinfo = -99;
        CHECK(lk_basis_create(ctx, LK_F64, n, nc, &XB));
        CHECK(lk_vec_rand(XB, 0, 21, 0, 0));
        CHECK(lk_vec_rand(XB, 1, 22, 0, 0));
        double R[4] = {0, 0, 0, 0};
        CHECK(lk_qr(XB, 0, p, R, p, 1e-15, &qinfo));
        EXPECT(qinfo == 0 && R[0] > 0.0 && R[3] > 0.0 && R[1] == 0.0);
        double *HB = calloc((size_t)nc * kd * p, sizeof(double)), *HB2 = calloc((size_t)nc * kd * p, sizeof(double));
        CHECK(lk_arnoldi_block(A, XB, HB, nc, p, 1, kd, 1e-15, 0, &binfo));
        EXPECT(binfo == 0);
        double *xb = malloc((size_t)n * nc * sizeof(double));
        CHECK(lk_basis_download(XB, 0, nc, xb, n));
        double borth = 0.0, bres = 0.0;
        for (int a = 0; a < nc; ++a)
            for (int b2 = 0; b2 <= a; ++b2) {
                double g = 0.0;
                for (int64_t i = 0; i < n; ++i) g += xb[(size_t)a * n + i] * xb[(size_t)b2 * n + i];
                g = fabs(g - (a == b2 ? 1.0 : 0.0));
                if (g > borth) borth = g;
            }
        for (int j = 0; j < kd * p; ++j)
            for (int64_t i = 0; i < n; i += 997) {           /* a sample of rows of A X(:, j) - X+ H(:, j) */
                double r = d[i] * xb[(size_t)j * n + i];
                for (int a = 0; a < nc; ++a) r -= xb[(size_t)a * n + i] * HB[(size_t)j * nc + a];
                if (fabs(r) > bres) bres = fabs(r);
            }
        EXPECT(borth < 1e-12 && bres < 1e-12);
        /* the same factorisation in two ranges from the same starting block */
        lk_basis_t XC;
        CHECK(lk_basis_create(ctx, LK_F64, n, nc, &XC));
        CHECK(lk_vec_rand(XC, 0, 21, 0, 0));
        CHECK(lk_vec_rand(XC, 1, 22, 0, 0));
        CHECK(lk_qr(XC, 0, p, R, p, 1e-15, &qinfo));
        CHECK(lk_arnoldi_block(A, XC, HB2, nc, p, 1, 2, 1e-15, 0, &binfo));
        CHECK(lk_arnoldi_block(A, XC, HB2, nc, p, 3, kd, 1e-15, 0, &binfo));
        for (int i = 0; i < nc * kd * p; ++i) EXPECT(HB[i] == HB2[i]);
        EXPECT(lk_arnoldi_block(A, XC, HB2, nc - 1, p, 1, kd, 1e-15, 0, &binfo) == LK_ERR_INVALID);   /* ldh too small */
        EXPECT(lk_qr(XC, nc - 1, p, R, p, 1e-15, &qinfo) == LK_ERR_INVALID);                          /* columns beyond the panel */
        int64_t rs3[3] = {-1, -1, -1};
        CHECK(lk_resident_stats(ctx, rs3));
        EXPECT(rs3[0] > 0 && rs3[1] == 0);                   /* these panels fit the caches: single launches ran, none gave up */
        EXPECT(lk_set_tuning(ctx, "xhy_debug", 1) == LK_ERR_INVALID);   /* wrong-result diagnostics are not in the shipped library */
        free(xb); free(HB); free(HB2);
        lk_basis_destroy(XB); lk_basis_destroy(XC);
    }

    /* error convention: negative status + message, nothing aborts */
    EXPECT(lk_vec_zero(X, m + 5) == LK_ERR_INVALID && strstr(lk_last_error(), "out of range"));
    EXPECT(lk_dgs(X, 3, X, 1, h, norms, 0, &dinfo) == LK_ERR_INVALID);   /* y among the basis columns */
    EXPECT(lk_basis_wrap(ctx, LK_F64, n, 2, n, (char *)dev + 8, &W) == LK_ERR_INVALID);  /* misaligned */
    lk_basis_t Z; EXPECT(lk_basis_create(ctx, 7, n, 1, &Z) == LK_ERR_INVALID);
    EXPECT(lk_arnoldi(A, X, H, m, 1, m, 1e-15, 0, &info) == LK_ERR_INVALID);  /* ldh too small */

    lk_basis_destroy(W); hipFree(dev);
    lk_linop_destroy(A); lk_basis_destroy(X); lk_finalize(ctx);
    printf("C client ok: orth=%.2e relation=%.2e\n", orth, res);
    return 0;
}
