!> tools/plugin_check/driver.f90 -- BUILD-CONTAINER check of fortran/dense_vector_gpu.f90 (see tools/check_plugin.sh).
!> Runs LightKrylov's OWN arnoldi / gmres / double_gram_schmidt_step on (a) the reference's dense_vector / dense_linop
!> and (b) the plugin's dense_vector_gpu / dense_linop_gpu, and compares; then exercises the object-semantics cases
!> (deep-copy assignment, bit copies, intent(out) re-acquisition, column re-use).  Linked twice by the script: against
!> liblightkrylov_hip.so (link check; cannot run without a GPU) and against the host mock of the C ABI (executed).
!> `eigs` is referenced so that it must link, but not executed (the stdlib stand-ins have no geev / trsen).
program plugin_driver
    use, intrinsic :: iso_c_binding
    use LightKrylov_Constants, only: dp
    use LightKrylov_Logger, only: logger_setup
    use LightKrylov_AbstractVectors
    use LightKrylov_AbstractLinops
    use LightKrylov_BaseKrylov, only: arnoldi, double_gram_schmidt_step
    use LightKrylov_IterativeSolvers, only: gmres, eigs, gmres_dp_opts
    use lightkrylov_gpu
    implicit none
    integer, parameter :: n = 120, m = 12
    integer :: nfail = 0

    call logger_setup(log_level=100, log_stdout=.false.)
    call lk_gpu_init(0)
    call check_arnoldi_rdp()
    call check_arnoldi_cdp()
    call check_gmres_rdp()
    call check_gmres_cdp()
    call check_assignment_is_deep()
    call check_pool_is_bounded()
    if (command_argument_count() > 99) call never_executed_eigs()
    call lk_gpu_release_all()
    call lk_gpu_finalize()
    if (nfail /= 0) then
        print '(A,I0,A)', 'plugin_driver: ', nfail, ' check(s) FAILED'
        error stop 1
    end if
    print '(A)', 'plugin_driver: all checks passed'

contains

    subroutine report(name, err, tol)
        character(len=*), intent(in) :: name
        real(dp), intent(in) :: err, tol
        if (err <= tol) then
            print '(A,A,ES10.2)', '  ok   ', name, err
        else
            print '(A,A,ES10.2,A,ES10.2)', '  FAIL ', name, err, ' > ', tol
            nfail = nfail + 1
        end if
    end subroutine

    subroutine test_matrix_rdp(A, x0)
        real(dp), intent(out) :: A(n, n), x0(n)
        integer :: i, j
        do j = 1, n
            do i = 1, n
                A(i, j) = sin(real(3*i + 7*j, dp))/real(n, dp)
            end do
            A(j, j) = A(j, j) + 1.0_dp + real(j, dp)/real(n, dp)
            x0(j) = cos(real(j, dp))
        end do
        x0 = x0/sqrt(sum(x0**2))
    end subroutine

    subroutine check_arnoldi_rdp()
        real(dp) :: A(n, n), x0(n), Href(m + 1, m), Hgpu(m + 1, m)
        type(dense_linop_rdp) :: Lref
        type(dense_linop_gpu_rdp) :: Lgpu
        type(dense_vector_rdp), allocatable :: Xref(:)
        type(dense_vector_gpu_rdp), allocatable :: Xgpu(:)
        type(dense_vector_gpu_rdp) :: b
        integer :: info, i
        integer(c_int64_t) :: st(4)
        call test_matrix_rdp(A, x0)
        Lref = dense_linop(A); Lgpu = dense_linop_gpu(A)
        allocate (Xref(m + 1)); do i = 1, m + 1; Xref(i)%n = n; call Xref(i)%zero(); end do
        Xref(1) = dense_vector(x0)
        Href = 0.0_dp; call arnoldi(Lref, Xref, Href, info)
        b%n = n; call b%zero()
        allocate (Xgpu(m + 1), source=b)              ! bit copies of b's handle, the reference's own idiom
        call zero_basis(Xgpu)
        call Xgpu(1)%upload(x0)
        Hgpu = 0.0_dp; call arnoldi(Lgpu, Xgpu, Hgpu, info)
        call report('arnoldi rdp: max |H_gpu - H_ref|            ', maxval(abs(Hgpu - Href)), 1.0e-12_dp)
        call lk_gpu_pool_stats(st)
        call report('arnoldi rdp: pool slabs (expect 1)           ', real(st(1), dp), 1.0_dp)
    end subroutine

    subroutine check_arnoldi_cdp()
        real(dp) :: Ar(n, n), x0r(n)
        complex(dp) :: A(n, n), x0(n), Href(m + 1, m), Hgpu(m + 1, m)
        type(dense_linop_cdp) :: Lref
        type(dense_linop_gpu_cdp) :: Lgpu
        type(dense_vector_cdp), allocatable :: Xref(:)
        type(dense_vector_gpu_cdp), allocatable :: Xgpu(:)
        type(dense_vector_gpu_cdp) :: b
        integer :: info, i
        call test_matrix_rdp(Ar, x0r)
        A = cmplx(Ar, 0.3_dp*transpose(Ar), kind=dp); x0 = cmplx(x0r, 0.5_dp*x0r(n:1:-1), kind=dp)
        x0 = x0/sqrt(sum(abs(x0)**2))
        Lref = dense_linop(A); Lgpu = dense_linop_gpu(A)
        allocate (Xref(m + 1)); do i = 1, m + 1; Xref(i)%n = n; call Xref(i)%zero(); end do
        Xref(1) = dense_vector(x0)
        Href = 0.0_dp; call arnoldi(Lref, Xref, Href, info)
        b%n = n; call b%zero()
        allocate (Xgpu(m + 1), source=b); call zero_basis(Xgpu)
        call Xgpu(1)%upload(x0)
        Hgpu = 0.0_dp; call arnoldi(Lgpu, Xgpu, Hgpu, info)
        call report('arnoldi cdp: max |H_gpu - H_ref|            ', maxval(abs(Hgpu - Href)), 1.0e-12_dp)
    end subroutine

    subroutine check_gmres_rdp()
        real(dp) :: A(n, n), rhs(n), xr(n), xg(n)
        type(dense_linop_rdp) :: Lref
        type(dense_linop_gpu_rdp) :: Lgpu
        type(dense_vector_rdp) :: bref, xref
        type(dense_vector_gpu_rdp) :: bgpu, xgpu
        type(gmres_dp_opts) :: opts
        integer :: info
        call test_matrix_rdp(A, rhs)
        Lref = dense_linop(A); Lgpu = dense_linop_gpu(A)
        opts = gmres_dp_opts(kdim=20, maxiter=5)
        bref = dense_vector(rhs); xref%n = n; call xref%zero()
        call gmres(Lref, bref, xref, info, rtol=1.0e-12_dp, atol=1.0e-14_dp, options=opts)
        xr = xref%data
        call bgpu%upload(rhs); xgpu%n = n; call xgpu%zero()
        call gmres(Lgpu, bgpu, xgpu, info, rtol=1.0e-12_dp, atol=1.0e-14_dp, options=opts)
        call xgpu%download(xg)
        call report('gmres rdp: max |x_gpu - x_ref|               ', maxval(abs(xg - xr)), 1.0e-11_dp)
        call report('gmres rdp: residual |A x - b|                ', maxval(abs(matmul(A, xg) - rhs)), 1.0e-10_dp)
    end subroutine

    subroutine check_gmres_cdp()
        real(dp) :: Ar(n, n), rr(n)
        complex(dp) :: A(n, n), rhs(n), xr(n), xg(n)
        type(dense_linop_cdp) :: Lref
        type(dense_linop_gpu_cdp) :: Lgpu
        type(dense_vector_cdp) :: bref, xref
        type(dense_vector_gpu_cdp) :: bgpu, xgpu
        type(gmres_dp_opts) :: opts
        integer :: info
        call test_matrix_rdp(Ar, rr)
        A = cmplx(Ar, 0.3_dp*transpose(Ar), kind=dp); rhs = cmplx(rr, rr(n:1:-1), kind=dp)
        Lref = dense_linop(A); Lgpu = dense_linop_gpu(A)
        opts = gmres_dp_opts(kdim=20, maxiter=5)
        bref = dense_vector(rhs); xref%n = n; call xref%zero()
        call gmres(Lref, bref, xref, info, rtol=1.0e-12_dp, atol=1.0e-14_dp, options=opts)
        xr = xref%data
        call bgpu%upload(rhs); xgpu%n = n; call xgpu%zero()
        call gmres(Lgpu, bgpu, xgpu, info, rtol=1.0e-12_dp, atol=1.0e-14_dp, options=opts)
        call xgpu%download(xg)
        call report('gmres cdp: max |x_gpu - x_ref|               ', maxval(abs(xg - xr)), 1.0e-11_dp)
    end subroutine

    !> CG's update pattern (`p = r`, then r changes, then p is used: CG.fypp:116,131,163) and friends
    subroutine check_assignment_is_deep()
        real(dp) :: x0(n), got(n)
        type(dense_vector_gpu_rdp) :: r
        class(abstract_vector_rdp), allocatable :: p, y
        integer :: i
        do i = 1, n; x0(i) = real(i, dp); end do
        call r%upload(x0)
        p = r                                   ! intrinsic polymorphic assignment -> handle's defined assignment
        call r%scal(2.0_dp)                     ! must not reach p
        select type (p); class is (dense_vector_gpu_rdp); call p%download(got); end select
        call report('p = r; r%scal(2): p unchanged                ', maxval(abs(got - x0)), 0.0_dp)
        call p%axpby(1.0_dp, r, -1.0_dp)        ! p = r - p = x0
        select type (p); class is (dense_vector_gpu_rdp); call p%download(got); end select
        call report('p%axpby(1, r, -1) = 2 x0 - x0                ', maxval(abs(got - x0)), 0.0_dp)
        allocate (y, source=r)                  ! bit copy: shares r's column until first written
        call y%scal(0.5_dp)                     ! copy-on-write: y gets its own column, r keeps 2 x0
        call r%download(got)
        call report('allocate(y, source=r); y%scal: r unchanged   ', maxval(abs(got - 2.0_dp*x0)), 0.0_dp)
        select type (y); class is (dense_vector_gpu_rdp); call y%download(got); end select
        call report('                             y = x0          ', maxval(abs(got - x0)), 0.0_dp)
        call copy(y, r)                         ! intent(out) dummy + axpby(1, from, 0)
        select type (y); class is (dense_vector_gpu_rdp); call y%download(got); end select
        call report('copy(y, r)                                   ', maxval(abs(got - 2.0_dp*x0)), 0.0_dp)
    end subroutine

    !> 200 Gram-Schmidt steps allocate 800 temporaries inside the reference (linear_combination's `y`, twice per
    !> pass); device memory must not grow with the number of calls.
    subroutine check_pool_is_bounded()
        real(dp) :: A(n, n), x0(n)
        type(dense_vector_gpu_rdp), allocatable :: X(:)
        type(dense_vector_gpu_rdp) :: b, y
        integer(c_int64_t) :: st0(4), st1(4)
        real(dp) :: beta(m)
        integer :: i, info
        call test_matrix_rdp(A, x0)
        b%n = n; call b%zero()
        allocate (X(m), source=b); call zero_basis(X)
        do i = 1, m; call X(i)%rand(.true.); end do
        call y%upload(x0)
        do i = 1, 5
            call double_gram_schmidt_step(y, X, info, if_chk_orthonormal=.false., beta=beta)
        end do
        call lk_gpu_pool_stats(st0)
        do i = 1, 200
            call y%upload(x0)
            call double_gram_schmidt_step(y, X, info, if_chk_orthonormal=.false., beta=beta)
        end do
        call lk_gpu_pool_stats(st1)
        print '(A,4I8)', '       pool after   5 DGS calls: slabs, carved, live, reused = ', st0
        print '(A,4I8)', '       pool after 205 DGS calls: slabs, carved, live, reused = ', st1
        call report('pool: columns carved by 200 more DGS calls   ', real(st1(2) - st0(2), dp), 4.0_dp)
        call report('pool: slabs                                  ', real(st1(1), dp), 3.0_dp)
    end subroutine

    subroutine never_executed_eigs()
        real(dp) :: A(n, n), x0(n)
        type(dense_linop_gpu_rdp) :: Lgpu
        type(dense_vector_gpu_rdp), allocatable :: X(:)
        complex(dp), allocatable :: lambda(:)
        real(dp), allocatable :: res(:)
        integer :: info
        call test_matrix_rdp(A, x0)
        Lgpu = dense_linop_gpu(A)
        allocate (X(4)); X%n = n
        call eigs(Lgpu, X, lambda, res, info, kdim=16)
    end subroutine
end program plugin_driver
