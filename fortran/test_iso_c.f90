!> Fortran host program driving the engine through ISO_C_BINDING only (no LightKrylov needed):
!> Arnoldi on the diagonal operator d_i = 1 + (i-1)/n, x0_i = sin(i)/||.||, n = 1000, m = 8 --
!> the case SURVEY.md Appendix A records for the reference's own arnoldi.  Prints H(1,1), H(2,1),
!> H(m+1,m) and max |X^T X - I| so tests/test_fortran_binding.py can compare with the oracle.
!> progress function for lk_arnoldi_segments (a bind(C) module procedure: c_funloc needs an interoperable procedure)
module test_iso_c_progress
    use, intrinsic :: iso_c_binding
    implicit none
    integer :: ncalls = 0, last_reported = 0
    logical :: in_order = .true.
contains
    function on_columns(user, kfirst, klast) bind(C) result(stop_now)
        type(c_ptr), value :: user
        integer(c_int), value :: kfirst, klast
        integer(c_int) :: stop_now
        if (kfirst /= last_reported + 1 .or. klast < kfirst) in_order = .false.
        last_reported = klast
        ncalls = ncalls + 1
        stop_now = 0_c_int
    end function
end module test_iso_c_progress

program test_iso_c
    use, intrinsic :: iso_c_binding
    use lightkrylov_hip_c
    use test_iso_c_progress
    implicit none
    integer, parameter :: n = 1000, m = 8
    type(c_ptr) :: ctx, X, A
    real(c_double), target :: d(n), x0(n), Xh(n, m + 1)
    real(c_double) :: H(m + 1, m), G, orth, nrm
    integer(c_int) :: rc, info
    integer :: i, j, k
    ! lazy per-object pass
    type(c_ptr) :: Z
    real(c_double) :: hl(m), hf(m), d2(2), ny_lazy, ny_fused, one(1), ai(1), mone(1)
    integer(c_int64_t) :: st(4), fs(4), ps0(4), ps1(4)
    ! complex(dp) pass + column pool
    type(c_ptr) :: Xz, Az, slab, slab2
    complex(c_double_complex), target :: dz(n), x0z(n)
    complex(c_double_complex) :: Hz(m + 1, m)
    integer(c_int) :: col2, cols(m + 1)
    integer(c_intptr_t) :: tag
    integer :: it

    rc = lk_init(0_c_int, c_null_ptr, ctx); call chk(rc, 'lk_init')
    do i = 1, n
        d(i) = 1.0d0 + real(i - 1, c_double)/real(n, c_double)
        x0(i) = sin(real(i, c_double))
    end do
    x0 = x0/sqrt(sum(x0**2))
    rc = lk_basis_create(ctx, LK_F64, int(n, c_int64_t), int(m + 1, c_int), X); call chk(rc, 'lk_basis_create')
    rc = lk_basis_upload(X, 0_c_int, 1_c_int, c_loc(x0), int(n, c_int64_t)); call chk(rc, 'lk_basis_upload')
    rc = lk_linop_diag_create(ctx, LK_F64, int(n, c_int64_t), c_loc(d), A); call chk(rc, 'lk_linop_diag_create')
    H = 0.0d0
    rc = lk_arnoldi(A, X, H, int(m + 1, c_int64_t), 1_c_int, int(m, c_int), 1.0d-15, 0_c_int, info)
    call chk(rc, 'lk_arnoldi')
    rc = lk_basis_download(X, 0_c_int, int(m + 1, c_int), c_loc(Xh), int(n, c_int64_t)); call chk(rc, 'lk_basis_download')
    orth = 0.0d0
    do j = 1, m + 1
        do k = 1, m + 1
            G = dot_product(Xh(:, j), Xh(:, k))
            if (j == k) G = G - 1.0d0
            orth = max(orth, abs(G))
        end do
    end do
    rc = lk_vec_norm(X, int(m, c_int), nrm); call chk(rc, 'lk_vec_norm')
    print '(A,I0)', 'info ', info
    print '(A,ES24.16)', 'H11 ', H(1, 1)
    print '(A,ES24.16)', 'H21 ', H(2, 1)
    print '(A,ES24.16)', 'Hlast ', H(m + 1, m)
    print '(A,ES12.4)', 'orth ', orth
    print '(A,ES24.16)', 'norm_last ', nrm
    ! ---- the same factorisation delivered in segments through a Fortran progress function (lk_arnoldi_segments, round 5)
    block
        type(c_ptr) :: X2
        real(c_double) :: H2(m + 1, m)
        integer(c_int) :: info2, segs(2)
        rc = lk_basis_create(ctx, LK_F64, int(n, c_int64_t), int(m + 1, c_int), X2); call chk(rc, 'lk_basis_create(X2)')
        rc = lk_basis_upload(X2, 0_c_int, 1_c_int, c_loc(x0), int(n, c_int64_t)); call chk(rc, 'lk_basis_upload(X2)')
        H2 = 0.0d0
        segs = [3_c_int, 6_c_int]
        rc = lk_arnoldi_segments(A, X2, H2, int(m + 1, c_int64_t), 1_c_int, int(m, c_int), 1.0d-15, 0_c_int, segs, 2_c_int, &
                                 c_funloc(on_columns), c_null_ptr, info2)
        call chk(rc, 'lk_arnoldi_segments')
        print '(A,I0)', 'seg_info ', info2
        print '(A,I0)', 'seg_calls ', ncalls
        print '(A,I0)', 'seg_last ', last_reported
        print '(A,I0)', 'seg_in_order ', merge(1, 0, in_order)
        print '(A,ES12.4)', 'seg_H_diff ', maxval(abs(H2 - H))
        rc = lk_basis_destroy(X2)
    end block
    ! ---- the reference's per-object schedule (innerprod loop, then linear_combination loop, then sub:
    !      gram_schmidt.fypp:141-145 through AbstractVectors.fypp:672-674, 600-602) driven from Fortran with
    !      the engine in lazy mode: m dots must cost ONE sweep; proj stays virtual and y%sub(proj) + the next norm ONE more.
    one = 1.0d0; mone = -1.0d0
    rc = lk_set_tuning(ctx, 'lazy'//c_null_char, 1_c_int); call chk(rc, 'lk_set_tuning')
    rc = lk_basis_create(ctx, LK_F64, int(n, c_int64_t), int(m + 3, c_int), Z); call chk(rc, 'lk_basis_create(Z)')
    do j = 0, m - 1
        rc = lk_vec_copy(Z, int(j, c_int), X, int(j, c_int)); call chk(rc, 'lk_vec_copy')
    end do
    rc = lk_vec_rand(Z, int(m, c_int), 99_c_int64_t, 0_c_int64_t, 0_c_int); call chk(rc, 'lk_vec_rand')      ! y
    rc = lk_vec_copy(Z, int(m + 2, c_int), Z, int(m, c_int)); call chk(rc, 'lk_vec_copy')                     ! y copy for the fused call
    do j = 0, m - 1                                                                                          ! innerprod
        rc = lk_vec_dot(Z, int(j, c_int), Z, int(m, c_int), d2); call chk(rc, 'lk_vec_dot')
        hl(j + 1) = d2(1)
    end do
    rc = lk_vec_zero(Z, int(m + 1, c_int)); call chk(rc, 'lk_vec_zero')                                      ! proj
    do j = 0, m - 1                                                                                          ! linear_combination
        ai(1) = hl(j + 1)
        rc = lk_vec_axpby(ai, Z, int(j, c_int), one, Z, int(m + 1, c_int)); call chk(rc, 'lk_vec_axpby')
    end do
    rc = lk_vec_axpby(mone, Z, int(m + 1, c_int), one, Z, int(m, c_int)); call chk(rc, 'lk_vec_axpby(sub)')  ! y%sub(proj)
    rc = lk_vec_norm(Z, int(m, c_int), ny_lazy); call chk(rc, 'lk_vec_norm')
    rc = lk_lazy_stats(ctx, st); call chk(rc, 'lk_lazy_stats')
    rc = lk_lazy_fusion_stats(ctx, fs); call chk(rc, 'lk_lazy_fusion_stats')
    rc = lk_orthogonalize(Z, int(m, c_int), Z, int(m + 2, c_int), hf, info); call chk(rc, 'lk_orthogonalize')
    rc = lk_vec_norm(Z, int(m + 2, c_int), ny_fused); call chk(rc, 'lk_vec_norm')
    print '(A,I0)', 'lazy_hits ', st(1)
    print '(A,I0)', 'lazy_sweeps ', st(2)
    print '(A,I0)', 'lazy_queued ', st(3)
    print '(A,I0)', 'lazy_flushes ', st(4)
    print '(A,I0)', 'lazy_fused_sweeps ', fs(1)
    print '(A,I0)', 'lazy_temporaries_written ', fs(4)
    print '(A,ES12.4)', 'lazy_h_err ', maxval(abs(hl - hf))
    print '(A,ES12.4)', 'lazy_y_err ', abs(ny_lazy - ny_fused)
    rc = lk_basis_destroy(Z)
    rc = lk_set_tuning(ctx, 'lazy'//c_null_char, 0_c_int); call chk(rc, 'lk_set_tuning')

    ! ---- complex(dp) kind through the same binding: Arnoldi with a complex diagonal operator (the python test
    !      recomputes this case with the oracle and compares the printed entries)
    do i = 1, n
        dz(i) = cmplx(1.0d0 + real(i - 1, c_double)/real(n, c_double), 0.25d0*sin(real(i, c_double)), kind=c_double_complex)
        x0z(i) = cmplx(sin(real(i, c_double)), cos(real(2*i, c_double)), kind=c_double_complex)
    end do
    x0z = x0z/sqrt(sum(abs(x0z)**2))
    rc = lk_basis_create(ctx, LK_C128, int(n, c_int64_t), int(m + 1, c_int), Xz); call chk(rc, 'lk_basis_create(z)')
    rc = lk_basis_upload(Xz, 0_c_int, 1_c_int, c_loc(x0z), int(n, c_int64_t)); call chk(rc, 'lk_basis_upload(z)')
    rc = lk_linop_diag_create(ctx, LK_C128, int(n, c_int64_t), c_loc(dz), Az); call chk(rc, 'lk_linop_diag_create(z)')
    Hz = (0.0d0, 0.0d0)
    call arnoldi_z(Az, Xz, Hz, info)
    print '(A,I0)', 'z_info ', info
    print '(A,ES24.16)', 'z_H11_re ', real(Hz(1, 1))
    print '(A,ES24.16)', 'z_H11_im ', aimag(Hz(1, 1))
    print '(A,ES24.16)', 'z_H12_re ', real(Hz(1, 2))
    print '(A,ES24.16)', 'z_H12_im ', aimag(Hz(1, 2))
    print '(A,ES24.16)', 'z_Hlast ', real(Hz(m + 1, m))
    rc = lk_linop_destroy(Az)
    rc = lk_basis_destroy(Xz)

    ! ---- the other fused factorisations through the same binding (round 6): lk_lanczos, lk_bidiag, lk_arnoldi_block -- every entry the
    !      python test compares with the oracle is printed as `<name>_<i>_<j> value`
    block
        type(c_ptr) :: XL, UB, VB, XK
        real(c_double) :: T(m + 1, m), Bd(m + 1, m)
        real(c_double), target :: x2(n, 2)
        real(c_double) :: HB(2*(m/2 + 1), 2*(m/2))
        integer(c_int) :: li
        rc = lk_basis_create(ctx, LK_F64, int(n, c_int64_t), int(m + 1, c_int), XL); call chk(rc, 'lk_basis_create(XL)')
        rc = lk_basis_upload(XL, 0_c_int, 1_c_int, c_loc(x0), int(n, c_int64_t)); call chk(rc, 'lk_basis_upload(XL)')
        T = 0.0d0
        rc = lk_lanczos(A, XL, T, int(m + 1, c_int64_t), 1_c_int, int(m, c_int), 1.0d-15, li); call chk(rc, 'lk_lanczos')
        print '(A,I0)', 'lz_info ', li
        do j = 1, m
            do i = 1, m + 1
                print '(A,I0,A,I0,ES25.16E3)', 'lz_T_', i, '_', j, T(i, j)
            end do
        end do
        rc = lk_basis_destroy(XL)
        rc = lk_basis_create(ctx, LK_F64, int(n, c_int64_t), int(m + 1, c_int), UB); call chk(rc, 'lk_basis_create(UB)')
        rc = lk_basis_create(ctx, LK_F64, int(n, c_int64_t), int(m, c_int), VB); call chk(rc, 'lk_basis_create(VB)')
        rc = lk_basis_upload(UB, 0_c_int, 1_c_int, c_loc(x0), int(n, c_int64_t)); call chk(rc, 'lk_basis_upload(UB)')
        Bd = 0.0d0
        rc = lk_bidiag(A, UB, VB, Bd, int(m + 1, c_int64_t), 1_c_int, int(m, c_int), 1.0d-15, li); call chk(rc, 'lk_bidiag')
        print '(A,I0)', 'bd_info ', li
        do j = 1, m
            do i = 1, m + 1
                print '(A,I0,A,I0,ES25.16E3)', 'bd_B_', i, '_', j, Bd(i, j)
            end do
        end do
        rc = lk_basis_destroy(UB); rc = lk_basis_destroy(VB)
        ! block Arnoldi, blksize = 2: starting block = [x0, cos(i)] orthonormalised by lk_qr on the device
        do i = 1, n
            x2(i, 1) = x0(i)
            x2(i, 2) = cos(real(i, c_double))
        end do
        rc = lk_basis_create(ctx, LK_F64, int(n, c_int64_t), int(2*(m/2 + 1), c_int), XK); call chk(rc, 'lk_basis_create(XK)')
        rc = lk_basis_upload(XK, 0_c_int, 2_c_int, c_loc(x2), int(n, c_int64_t)); call chk(rc, 'lk_basis_upload(XK)')
        block
            real(c_double) :: R2(2, 2)
            rc = lk_qr(XK, 0_c_int, 2_c_int, R2, 2_c_int64_t, 1.0d-15, li); call chk(rc, 'lk_qr')
            print '(A,I0)', 'qr_info ', li
            print '(A,ES25.16E3)', 'qr_R11 ', R2(1, 1)
            print '(A,ES25.16E3)', 'qr_R12 ', R2(1, 2)
            print '(A,ES25.16E3)', 'qr_R22 ', R2(2, 2)
        end block
        HB = 0.0d0
        rc = lk_arnoldi_block(A, XK, HB, int(size(HB, 1), c_int64_t), 2_c_int, 1_c_int, int(m/2, c_int), 1.0d-15, 0_c_int, li)
        call chk(rc, 'lk_arnoldi_block')
        print '(A,I0)', 'bk_info ', li
        do j = 1, size(HB, 2)
            do i = 1, size(HB, 1)
                print '(A,I0,A,I0,ES25.16E3)', 'bk_H_', i, '_', j, HB(i, j)
            end do
        end do
        rc = lk_basis_destroy(XK)
    end block

    ! ---- column pool, driven the way the LightKrylov plugin drives it (fortran/dense_vector_gpu.f90): V(1..m+1)
    !      acquired in order land in consecutive columns of ONE slab; 200 emulated Gram-Schmidt passes, each of which
    !      "allocates" its two temporaries (linear_combination's y, AbstractVectors.fypp:595-598) at two recurring
    !      addresses, must not carve a single new column.
    do j = 1, m + 1
        tag = int(4096 + 64*j, c_intptr_t)
        rc = lk_pool_acquire(ctx, LK_F64, int(n, c_int64_t), tag, slab, cols(j)); call chk(rc, 'lk_pool_acquire')
        rc = lk_vec_rand(slab, cols(j), int(500 + j, c_int64_t), 0_c_int64_t, 1_c_int); call chk(rc, 'lk_vec_rand(pool)')
    end do
    rc = lk_pool_stats(ctx, ps0); call chk(rc, 'lk_pool_stats')
    one = 1.0d0; mone = -1.0d0
    do it = 1, 200
        tag = int(900000 + 64*mod(it, 2), c_intptr_t)                     ! proj of pass 1 / pass 2
        rc = lk_pool_acquire(ctx, LK_F64, int(n, c_int64_t), tag, slab2, col2); call chk(rc, 'lk_pool_acquire(proj)')
        rc = lk_vec_zero(slab2, col2); call chk(rc, 'lk_vec_zero(proj)')
        do j = 1, m
            rc = lk_vec_dot(slab, cols(j), slab, cols(m + 1), d2); call chk(rc, 'lk_vec_dot(pool)')
            ai(1) = d2(1)
            rc = lk_vec_axpby(ai, slab, cols(j), one, slab2, col2); call chk(rc, 'lk_vec_axpby(pool)')
        end do
        rc = lk_vec_axpby(mone, slab2, col2, one, slab, cols(m + 1)); call chk(rc, 'lk_vec_axpby(pool sub)')
    end do
    rc = lk_pool_stats(ctx, ps1); call chk(rc, 'lk_pool_stats')
    print '(A,I0)', 'pool_consecutive ', merge(1, 0, all(cols == [(j - 1, j=1, m + 1)]))
    print '(A,I0)', 'pool_slabs ', ps1(1)
    print '(A,I0)', 'pool_carved_before ', ps0(2)
    print '(A,I0)', 'pool_carved_after ', ps1(2)
    print '(A,I0)', 'pool_reused ', ps1(4)
    rc = lk_vec_dot(slab, cols(1), slab, cols(m + 1), d2); call chk(rc, 'lk_vec_dot(final)')
    print '(A,ES12.4)', 'pool_orth_resid ', abs(d2(1))
    rc = lk_pool_release_all(ctx); call chk(rc, 'lk_pool_release_all')
    rc = lk_linop_destroy(A)
    rc = lk_basis_destroy(X)
    rc = lk_finalize(ctx)
contains
    !> lk_arnoldi takes the Hessenberg array as interleaved doubles; a complex(dp) Fortran array IS that layout
    subroutine arnoldi_z(Aop, Xb, Hc, inf)
        type(c_ptr), intent(in) :: Aop, Xb
        complex(c_double_complex), intent(inout), target :: Hc(:, :)
        integer(c_int), intent(out) :: inf
        real(c_double), pointer :: Hr(:)
        call c_f_pointer(c_loc(Hc), Hr, [2*size(Hc)])
        rc = lk_arnoldi(Aop, Xb, Hr, int(size(Hc, 1), c_int64_t), 1_c_int, int(size(Hc, 2), c_int), 1.0d-15, 0_c_int, inf)
        call chk(rc, 'lk_arnoldi(z)')
    end subroutine
    subroutine chk(rc, what)
        integer(c_int), intent(in) :: rc
        character(len=*), intent(in) :: what
        if (rc /= LK_OK) then
            print '(A,A,A,I0,A,A)', 'ERROR in ', what, ': rc=', rc, ' ', lk_error_message()
            stop 1
        end if
    end subroutine
end program test_iso_c
