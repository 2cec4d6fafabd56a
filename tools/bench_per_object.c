/* The reference's arnoldi as a COMPILED host drives the plugin: the exact per-object call sequence of
 * arnoldi.fypp:34-73 -> gram_schmidt.fypp:12-57, 113-154 -> AbstractVectors.fypp:571-603 through the C ABI (what
 * fortran/dense_vector_gpu.f90 issues: norm as dot(y, y), k dots, pool-acquire + zero + k axpbys for proj, y%sub(proj),
 * twice; then qr's norm and scal), with no interpreter in between -- eager vs lazy engine vs the fused lk_arnoldi.
 *   gcc -O2 -o bench_per_object tools/bench_per_object.c -Iinclude -Llightkrylov_amd -llightkrylov_hip -lm -Wl,-rpath,$PWD/lightkrylov_amd
 *   ./bench_per_object [rows=10000000] [m=64] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "../include/lightkrylov_hip.h"

#define CK(call) do { int rc_ = (call); if (rc_ != LK_OK) { printf("FAIL %s: %s\n", #call, lk_last_error()); exit(1); } } while (0)
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* one Gram-Schmidt pass the way orthogonalize_vector_against_basis issues it */
static void cgs_pass(lk_context_t ctx, lk_basis_t X, int k, int jy, int64_t n, double *h) {
    double d[2], one = 1.0, mone = -1.0;
    lk_basis_t slab; int col;
    CK(lk_vec_dot(X, jy, X, jy, d));                                     /* y%norm() < atol ? */
    for (int i = 0; i < k; ++i) { CK(lk_vec_dot(X, i, X, jy, d)); h[i] = d[0]; }        /* innerprod */
    CK(lk_pool_acquire(ctx, LK_F64, n, 0xC0FFEEull, &slab, &col));       /* allocate(proj, source=X(1)): same address every call */
    CK(lk_vec_zero(slab, col));
    for (int i = 0; i < k; ++i) CK(lk_vec_axpby(&h[i], X, i, &one, slab, col));         /* linear_combination */
    CK(lk_vec_axpby(&mone, slab, col, &one, X, jy));                     /* y%sub(proj) */
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int m = argc > 2 ? atoi(argv[2]) : 64;
    const char *modes[3] = {"eager", "lazy", "fused"};
    double alg = 0; for (int k = 1; k <= m; ++k) alg += 8.0 * n * (3.0 * k + 5.0);
    printf("{\"n\": %lld, \"m\": %d", (long long)n, m);
    for (int mode = 0; mode < 3; ++mode) {
        lk_context_t ctx; lk_basis_t X; lk_linop_t A;
        CK(lk_init(0, NULL, &ctx));
        CK(lk_set_tuning(ctx, "lazy", mode == 1));
        CK(lk_basis_create(ctx, LK_F64, n, m + 1, &X));
        CK(lk_linop_diag_linspace_create(ctx, n, 0, 1.0, 1.0 / (double)n, &A));
        double *H = calloc((size_t)(m + 1) * m, sizeof(double)), *h1 = malloc(m * sizeof(double)), *h2 = malloc(m * sizeof(double));
        double best = 1e30;
        for (int rep = 0; rep < 2; ++rep) {
            CK(lk_vec_rand(X, 0, 7, 0, 1));
            CK(lk_sync(ctx));
            const double t0 = now();
            if (mode == 2) {
                int info; CK(lk_arnoldi(A, X, H, m + 1, 1, m, 1e-15, 0, &info));
            } else {
                for (int k = 1; k <= m; ++k) {
                    double d[2], s;
                    CK(lk_linop_apply(A, LK_OP_N, X, k - 1, X, k));
                    cgs_pass(ctx, X, k, k, n, h1);
                    cgs_pass(ctx, X, k, k, n, h2);
                    for (int i = 0; i < k; ++i) H[(size_t)(k - 1) * (m + 1) + i] = h1[i] + h2[i];
                    CK(lk_vec_dot(X, k, X, k, d));                       /* qr: beta = norm */
                    s = 1.0 / sqrt(d[0]); H[(size_t)(k - 1) * (m + 1) + k] = sqrt(d[0]);
                    CK(lk_vec_scal(X, k, &s));
                }
            }
            CK(lk_sync(ctx));
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        double fro = 0; for (size_t i = 0; i < (size_t)(m + 1) * m; ++i) fro += H[i] * H[i];
        printf(", \"%s\": {\"seconds\": %.6f, \"iters_per_s\": %.2f, \"GBps_on_algorithmic_3k+5\": %.0f, \"H_fro\": %.15g}", modes[mode], best,
               m / best, alg / best / 1e9, sqrt(fro));
        free(H); free(h1); free(h2);
        CK(lk_linop_destroy(A)); CK(lk_basis_destroy(X)); CK(lk_finalize(ctx));
    }
    printf("}\n");
    return 0;
}
