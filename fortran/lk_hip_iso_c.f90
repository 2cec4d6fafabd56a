!> ISO_C_BINDING interface to the MI355X Krylov engine (include/lightkrylov_hip.h).
!> One `bind(C)` interface per C entry point, same names, same argument order.
!> This module has no dependency on LightKrylov: it is the layer a LightKrylov plugin
!> (fortran/dense_vector_gpu.f90) and any other Fortran host code build on.
module lightkrylov_hip_c
    use, intrinsic :: iso_c_binding
    implicit none
    public

    integer(c_int), parameter :: LK_F64 = 0, LK_C128 = 1
    integer(c_int), parameter :: LK_OK = 0
    integer(c_int), parameter :: LK_DGS_NORMALIZE = 1
    integer(c_int), parameter :: LK_OP_N = 0, LK_OP_H = 1
    integer(c_int), parameter :: LK_ACCESS_READ = 0, LK_ACCESS_OVERWRITE = 1, LK_ACCESS_READWRITE = 2

    interface
        function lk_version() bind(C, name="lk_version") result(v)
            import :: c_int
            integer(c_int) :: v
        end function
        function lk_last_error() bind(C, name="lk_last_error") result(msg)
            import :: c_ptr
            type(c_ptr) :: msg
        end function
        function lk_init(device, stream, ctx) bind(C, name="lk_init") result(rc)
            import :: c_int, c_ptr
            integer(c_int), value :: device
            type(c_ptr), value :: stream
            type(c_ptr), intent(out) :: ctx
            integer(c_int) :: rc
        end function
        function lk_finalize(ctx) bind(C, name="lk_finalize") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int) :: rc
        end function
        function lk_sync(ctx) bind(C, name="lk_sync") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int) :: rc
        end function
        function lk_set_allreduce(ctx, fn, user, nranks, rank) bind(C, name="lk_set_allreduce") result(rc)
            import :: c_int, c_ptr, c_funptr
            type(c_ptr), value :: ctx, user
            type(c_funptr), value :: fn
            integer(c_int), value :: nranks, rank
            integer(c_int) :: rc
        end function
        function lk_set_partition(ctx, row0, n_global) bind(C, name="lk_set_partition") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), value :: row0, n_global
            integer(c_int) :: rc
        end function
        function lk_set_tuning(ctx, key, val) bind(C, name="lk_set_tuning") result(rc)
            import :: c_int, c_ptr, c_char
            type(c_ptr), value :: ctx
            character(kind=c_char), intent(in) :: key(*)
            integer(c_int), value :: val
            integer(c_int) :: rc
        end function
        function lk_lazy_stats(ctx, out4) bind(C, name="lk_lazy_stats") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), intent(out) :: out4(4)
            integer(c_int) :: rc
        end function
        function lk_lazy_speculation_stats(ctx, out2) bind(C, name="lk_lazy_speculation_stats") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), intent(out) :: out2(2)
            integer(c_int) :: rc
        end function
        function lk_resident_stats(ctx, out3) bind(C, name="lk_resident_stats") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), intent(out) :: out3(3)
            integer(c_int) :: rc
        end function
        function lk_resident_phase_ticks(ctx, out8) bind(C, name="lk_resident_phase_ticks") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), intent(out) :: out8(8)
            integer(c_int) :: rc
        end function
        function lk_lazy_fusion_stats(ctx, out4) bind(C, name="lk_lazy_fusion_stats") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), intent(out) :: out4(4)
            integer(c_int) :: rc
        end function
        function lk_orthogonalize(Bx, k, By, jy, h, info) bind(C, name="lk_orthogonalize") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: Bx, By
            integer(c_int), value :: k, jy
            real(c_double), intent(out) :: h(*)
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        function lk_basis_create(ctx, dtype, n_local, ncols, B) bind(C, name="lk_basis_create") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int), value :: dtype, ncols
            integer(c_int64_t), value :: n_local
            type(c_ptr), intent(out) :: B
            integer(c_int) :: rc
        end function
        function lk_basis_destroy(B) bind(C, name="lk_basis_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: B
            integer(c_int) :: rc
        end function
        function lk_basis_upload(B, col0, ncols, host, ldh) bind(C, name="lk_basis_upload") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: B, host
            integer(c_int), value :: col0, ncols
            integer(c_int64_t), value :: ldh
            integer(c_int) :: rc
        end function
        function lk_basis_download(B, col0, ncols, host, ldh) bind(C, name="lk_basis_download") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: B, host
            integer(c_int), value :: col0, ncols
            integer(c_int64_t), value :: ldh
            integer(c_int) :: rc
        end function
        function lk_vec_zero(B, j) bind(C, name="lk_vec_zero") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: B
            integer(c_int), value :: j
            integer(c_int) :: rc
        end function
        function lk_vec_rand(B, j, seed, row0, ifnorm) bind(C, name="lk_vec_rand") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: B
            integer(c_int), value :: j, ifnorm
            integer(c_int64_t), value :: seed, row0
            integer(c_int) :: rc
        end function
        function lk_vec_scal(B, j, alpha) bind(C, name="lk_vec_scal") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: B
            integer(c_int), value :: j
            real(c_double), intent(in) :: alpha(*)
            integer(c_int) :: rc
        end function
        function lk_vec_axpby(alpha, Bx, jx, beta, By, jy) bind(C, name="lk_vec_axpby") result(rc)
            import :: c_int, c_ptr, c_double
            real(c_double), intent(in) :: alpha(*), beta(*)
            type(c_ptr), value :: Bx, By
            integer(c_int), value :: jx, jy
            integer(c_int) :: rc
        end function
        function lk_vec_dot(Bx, jx, By, jy, res) bind(C, name="lk_vec_dot") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: Bx, By
            integer(c_int), value :: jx, jy
            real(c_double), intent(out) :: res(*)
            integer(c_int) :: rc
        end function
        function lk_vec_norm(B, j, res) bind(C, name="lk_vec_norm") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: B
            integer(c_int), value :: j
            real(c_double), intent(out) :: res
            integer(c_int) :: rc
        end function
        function lk_vec_copy(Bd, jd, Bs, js) bind(C, name="lk_vec_copy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: Bd, Bs
            integer(c_int), value :: jd, js
            integer(c_int) :: rc
        end function
        function lk_innerprod(Bx, k, By, jy0, p, M) bind(C, name="lk_innerprod") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: Bx, By
            integer(c_int), value :: k, jy0, p
            real(c_double), intent(out) :: M(*)
            integer(c_int) :: rc
        end function
        function lk_lincomb(Bx, k, C, q, By, jy0) bind(C, name="lk_lincomb") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: Bx, By
            integer(c_int), value :: k, q, jy0
            real(c_double), intent(in) :: C(*)
            integer(c_int) :: rc
        end function
        function lk_dgs(Bx, k, By, jy, h, norms, flags, info) bind(C, name="lk_dgs") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: Bx, By
            integer(c_int), value :: k, jy, flags
            real(c_double), intent(out) :: h(*), norms(3)
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        function lk_linop_diag_create(ctx, dtype, n_local, d_host, op) bind(C, name="lk_linop_diag_create") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx, d_host
            integer(c_int), value :: dtype
            integer(c_int64_t), value :: n_local
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_linop_dense_create(ctx, dtype, n, A_host, lda, op) bind(C, name="lk_linop_dense_create") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx, A_host
            integer(c_int), value :: dtype
            integer(c_int64_t), value :: n, lda
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_vec_device_ptr(B, j, access, dev_ptr) bind(C, name="lk_vec_device_ptr") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: B
            integer(c_int), value :: j, access
            type(c_ptr), intent(out) :: dev_ptr
            integer(c_int) :: rc
        end function
        function lk_linop_csr_create(ctx, dtype, n, rowptr, colind, vals, op) bind(C, name="lk_linop_csr_create") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx, rowptr, colind, vals
            integer(c_int), value :: dtype
            integer(c_int64_t), value :: n
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_context_info(ctx, device, stream) bind(C, name="lk_context_info") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), intent(out) :: device
            type(c_ptr), intent(out) :: stream
            integer(c_int) :: rc
        end function
        function lk_comm_info(ctx, nranks, rank) bind(C, name="lk_comm_info") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int), intent(out) :: nranks, rank
            integer(c_int) :: rc
        end function
        !> native RCCL all-reduce: rank 0 fills id(128) and ships it (e.g. MPI_Bcast); every rank then calls
        !> lk_comm_init_rank (collective).  lk_comm_available: local check that librccl resolves (agree on it over MPI
        !> before anyone enters the collective).
        function lk_comm_available() bind(C, name="lk_comm_available") result(rc)
            import :: c_int
            integer(c_int) :: rc
        end function
        function lk_comm_get_unique_id(id) bind(C, name="lk_comm_get_unique_id") result(rc)
            import :: c_int, c_char
            character(kind=c_char), intent(out) :: id(128)
            integer(c_int) :: rc
        end function
        function lk_comm_init_rank(ctx, nranks, rank, id) bind(C, name="lk_comm_init_rank") result(rc)
            import :: c_int, c_ptr, c_char
            type(c_ptr), value :: ctx
            integer(c_int), value :: nranks, rank
            character(kind=c_char), intent(in) :: id(128)
            integer(c_int) :: rc
        end function
        function lk_comm_destroy(ctx) bind(C, name="lk_comm_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int) :: rc
        end function
        function lk_pool_acquire(ctx, dtype, n_local, owner_tag, slab, col) bind(C, name="lk_pool_acquire") result(rc)
            import :: c_int, c_ptr, c_int64_t, c_intptr_t
            type(c_ptr), value :: ctx
            integer(c_int), value :: dtype
            integer(c_int64_t), value :: n_local
            integer(c_intptr_t), value :: owner_tag
            type(c_ptr), intent(out) :: slab
            integer(c_int), intent(out) :: col
            integer(c_int) :: rc
        end function
        function lk_pool_owner(ctx, slab, col, owner_tag) bind(C, name="lk_pool_owner") result(rc)
            import :: c_int, c_ptr, c_intptr_t
            type(c_ptr), value :: ctx, slab
            integer(c_int), value :: col
            integer(c_intptr_t), intent(out) :: owner_tag
            integer(c_int) :: rc
        end function
        function lk_pool_column_info(ctx, slab, col, owner_tag, generation) bind(C, name="lk_pool_column_info") result(rc)
            import :: c_int, c_ptr, c_intptr_t, c_int64_t
            type(c_ptr), value :: ctx, slab
            integer(c_int), value :: col
            integer(c_intptr_t), intent(out) :: owner_tag
            integer(c_int64_t), intent(out) :: generation
            integer(c_int) :: rc
        end function
        function lk_pool_release(ctx, slab, col) bind(C, name="lk_pool_release") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx, slab
            integer(c_int), value :: col
            integer(c_int) :: rc
        end function
        function lk_pool_release_all(ctx) bind(C, name="lk_pool_release_all") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: ctx
            integer(c_int) :: rc
        end function
        function lk_pool_stats(ctx, out4) bind(C, name="lk_pool_stats") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), intent(out) :: out4(4)
            integer(c_int) :: rc
        end function
        function lk_vec_size(B, n_local) bind(C, name="lk_vec_size") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: B
            integer(c_int64_t), intent(out) :: n_local
            integer(c_int) :: rc
        end function
        function lk_gram(Bx, k, G) bind(C, name="lk_gram") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: Bx
            integer(c_int), value :: k
            real(c_double), intent(out) :: G(*)
            integer(c_int) :: rc
        end function
        function lk_dgs_block(Bx, k, By, jy0, p, h, info) bind(C, name="lk_dgs_block") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: Bx, By
            integer(c_int), value :: k, jy0, p
            real(c_double), intent(out) :: h(*)
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        function lk_linop_diag_linspace_create(ctx, n_local, row0, d0, dstep, op) &
            bind(C, name="lk_linop_diag_linspace_create") result(rc)
            import :: c_int, c_ptr, c_int64_t, c_double
            type(c_ptr), value :: ctx
            integer(c_int64_t), value :: n_local, row0
            real(c_double), value :: d0, dstep
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_linop_lap5_create_sharded(ctx, N, j0, nj, op) bind(C, name="lk_linop_lap5_create_sharded") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), value :: N, j0, nj
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_linop_gl_create_sharded(ctx, n_global, row0, n_local, dx, tau, nsub, nu, gamma, mu_c, mu2, op) &
            bind(C, name="lk_linop_gl_create_sharded") result(rc)
            import :: c_int, c_ptr, c_int64_t, c_double
            type(c_ptr), value :: ctx
            integer(c_int64_t), value :: n_global, row0, n_local
            real(c_double), value :: dx, tau, mu_c, mu2
            integer(c_int), value :: nsub
            real(c_double), intent(in) :: nu(2), gamma(2)
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_set_halo_exchange(ctx, fn, user) bind(C, name="lk_set_halo_exchange") result(rc)
            import :: c_int, c_ptr, c_funptr
            type(c_ptr), value :: ctx, user
            type(c_funptr), value :: fn
            integer(c_int) :: rc
        end function
        !> all-gather of row blocks for the row-sharded dense / CSR operators (lk_allgather_fn in the header); a host with MPI
        !> passes c_funloc of a bind(C) wrapper around MPI_Allgatherv, lk_comm_init_rank installs the native RCCL one
        function lk_set_allgather(ctx, fn, user) bind(C, name="lk_set_allgather") result(rc)
            import :: c_int, c_ptr, c_funptr
            type(c_ptr), value :: ctx, user
            type(c_funptr), value :: fn
            integer(c_int) :: rc
        end function
        !> row-sharded dense_linop: row_starts(0:nranks) (the same on every rank), A_rows = this rank's n_local x n_global block
        function lk_linop_dense_create_sharded(ctx, dtype, n_global, row_starts, A_rows, lda, op) &
            bind(C, name="lk_linop_dense_create_sharded") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx, A_rows
            integer(c_int), value :: dtype
            integer(c_int64_t), value :: n_global, lda
            integer(c_int64_t), intent(in) :: row_starts(*)
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_linop_dense_wrap_sharded(ctx, dtype, n_global, row_starts, dev_ptr, lda, op) &
            bind(C, name="lk_linop_dense_wrap_sharded") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx, dev_ptr
            integer(c_int), value :: dtype
            integer(c_int64_t), value :: n_global, lda
            integer(c_int64_t), intent(in) :: row_starts(*)
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        !> row-sharded CSR operator: this rank's rows with GLOBAL 0-based column indices
        function lk_linop_csr_create_sharded(ctx, dtype, n_global, row_starts, rowptr, colind, vals, op) &
            bind(C, name="lk_linop_csr_create_sharded") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx, rowptr, colind, vals
            integer(c_int), value :: dtype
            integer(c_int64_t), value :: n_global
            integer(c_int64_t), intent(in) :: row_starts(*)
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_linop_lap5_create(ctx, N, op) bind(C, name="lk_linop_lap5_create") result(rc)
            import :: c_int, c_ptr, c_int64_t
            type(c_ptr), value :: ctx
            integer(c_int64_t), value :: N
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_linop_gl_create(ctx, n, dx, tau, nsub, nu, gamma, mu_c, mu2, op) bind(C, name="lk_linop_gl_create") result(rc)
            import :: c_int, c_ptr, c_int64_t, c_double
            type(c_ptr), value :: ctx
            integer(c_int64_t), value :: n
            real(c_double), value :: dx, tau, mu_c, mu2
            integer(c_int), value :: nsub
            real(c_double), intent(in) :: nu(2), gamma(2)
            type(c_ptr), intent(out) :: op
            integer(c_int) :: rc
        end function
        function lk_linop_destroy(op) bind(C, name="lk_linop_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: op
            integer(c_int) :: rc
        end function
        function lk_linop_apply(op, trans, Bx, jx, By, jy) bind(C, name="lk_linop_apply") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: op, Bx, By
            integer(c_int), value :: trans, jx, jy
            integer(c_int) :: rc
        end function
        function lk_arnoldi(A, X, H, ldh, kstart, kend, tol, trans, info) bind(C, name="lk_arnoldi") result(rc)
            import :: c_int, c_ptr, c_double, c_int64_t
            type(c_ptr), value :: A, X
            real(c_double), intent(inout) :: H(*)
            integer(c_int64_t), value :: ldh
            integer(c_int), value :: kstart, kend, trans
            real(c_double), value :: tol
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        !> lk_arnoldi delivered in segments while it runs: fn(user, kfirst, klast) (bind(C), returns c_int: non-zero = stop) is called
        !> as soon as columns kfirst..klast of H are final; seg_last(1:nseg) = last step of each segment (ascending)
        function lk_arnoldi_segments(A, X, H, ldh, kstart, kend, tol, trans, seg_last, nseg, fn, user, info) &
            bind(C, name="lk_arnoldi_segments") result(rc)
            import :: c_int, c_ptr, c_funptr, c_double, c_int64_t
            type(c_ptr), value :: A, X, user
            real(c_double), intent(inout) :: H(*)
            integer(c_int64_t), value :: ldh
            integer(c_int), value :: kstart, kend, trans, nseg
            real(c_double), value :: tol
            integer(c_int), intent(in) :: seg_last(*)
            type(c_funptr), value :: fn
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        function lk_bidiag(A, U, V, B, ldb, kstart, kend, tol, info) bind(C, name="lk_bidiag") result(rc)
            import :: c_int, c_ptr, c_double, c_int64_t
            type(c_ptr), value :: A, U, V
            real(c_double), intent(inout) :: B(*)
            integer(c_int64_t), value :: ldb
            integer(c_int), value :: kstart, kend
            real(c_double), value :: tol
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        function lk_qr(Q, j0, p, R, ldr, tol, info) bind(C, name="lk_qr") result(rc)
            import :: c_int, c_ptr, c_double, c_int64_t
            type(c_ptr), value :: Q
            integer(c_int), value :: j0, p
            real(c_double), intent(inout) :: R(*)
            integer(c_int64_t), value :: ldr
            real(c_double), value :: tol
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        function lk_arnoldi_block(A, X, H, ldh, blksize, kstart, kend, tol, trans, info) bind(C, name="lk_arnoldi_block") result(rc)
            import :: c_int, c_ptr, c_double, c_int64_t
            type(c_ptr), value :: A, X
            real(c_double), intent(inout) :: H(*)
            integer(c_int64_t), value :: ldh
            integer(c_int), value :: blksize, kstart, kend
            real(c_double), value :: tol
            integer(c_int), value :: trans
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
        function lk_lanczos(A, X, T, ldt, kstart, kend, tol, info) bind(C, name="lk_lanczos") result(rc)
            import :: c_int, c_ptr, c_double, c_int64_t
            type(c_ptr), value :: A, X
            real(c_double), intent(inout) :: T(*)
            integer(c_int64_t), value :: ldt
            integer(c_int), value :: kstart, kend
            real(c_double), value :: tol
            integer(c_int), intent(out) :: info
            integer(c_int) :: rc
        end function
    end interface

contains

    !> lk_last_error() as a Fortran string.
    function lk_error_message() result(msg)
        character(len=:), allocatable :: msg
        character(kind=c_char), pointer :: p(:)
        type(c_ptr) :: cp
        integer :: i, n
        cp = lk_last_error()
        msg = ''
        if (.not. c_associated(cp)) return
        call c_f_pointer(cp, p, [512])
        n = 0
        do i = 1, 512
            if (p(i) == c_null_char) exit
            n = i
        end do
        allocate (character(len=n) :: msg)
        do i = 1, n
            msg(i:i) = p(i)
        end do
    end function
end module lightkrylov_hip_c
