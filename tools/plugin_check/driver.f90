!> tools/plugin_check/driver.f90 -- BUILD-CONTAINER check of fortran/dense_vector_gpu.f90 (see tools/check_plugin.sh).
!> Runs LightKrylov's OWN arnoldi / gmres / double_gram_schmidt_step on (a) the reference's dense_vector / dense_linop
!> and (b) the plugin's dense_vector_gpu / dense_linop_gpu, and compares; then exercises the object-semantics cases
!> (deep-copy assignment, bit copies, intent(out) re-acquisition, column re-use).  Linked twice by the script: against
!> liblightkrylov_hip.so (link check; cannot run without a GPU) and against the host mock of the C ABI (executed).
!> `eigs` is referenced so that it must link, but not executed (the stdlib stand-ins have no geev / trsen).
!> A user operator on the REFERENCE's own dense_vector_rdp (a diagonal matrix), for the Appendix-A reproduction below.
module plugin_check_ref_ops
    use LightKrylov_Constants, only: dp
    use LightKrylov_AbstractVectors
    use LightKrylov_AbstractLinops
    implicit none
    type, extends(abstract_linop_rdp) :: diag_linop_ref
        real(dp), allocatable :: d(:)
    contains
        procedure, pass(self) :: matvec => ref_diag_matvec
        procedure, pass(self) :: rmatvec => ref_diag_matvec
    end type
contains
    subroutine ref_diag_matvec(self, vec_in, vec_out)
        class(diag_linop_ref), intent(inout) :: self
        class(abstract_vector_rdp), intent(in) :: vec_in
        class(abstract_vector_rdp), intent(out) :: vec_out
        select type (vec_in)
        type is (dense_vector_rdp)
            select type (vec_out)
            type is (dense_vector_rdp)
                vec_out%n = vec_in%n
                vec_out%data = self%d*vec_in%data
            end select
        end select
    end subroutine
end module plugin_check_ref_ops

program plugin_driver
    use, intrinsic :: iso_c_binding
    use LightKrylov_Constants, only: dp
    use LightKrylov_Logger, only: logger_setup
    use LightKrylov_AbstractVectors
    use LightKrylov_AbstractLinops
    use LightKrylov_BaseKrylov, only: arnoldi, double_gram_schmidt_step
    use LightKrylov_IterativeSolvers, only: gmres, eigs, cg, eighs, svds, gmres_dp_opts, gmres_dp_metadata, cg_dp_opts, cg_dp_metadata
    use stdlib_linalg, only: eigh, svd
    use lightkrylov_gpu
    use plugin_check_ref_ops
    implicit none
    integer, parameter :: n = 120, m = 12
    integer :: nfail = 0

    call logger_setup(log_level=100, log_stdout=.false.)
    call lk_gpu_init(0)
    call check_survey_values()
    call dump_reference_runs()
    call check_arnoldi_rdp()
    call check_arnoldi_cdp()
    call check_gmres_rdp()
    call check_gmres_cdp()
    call check_cg_rdp()
    call check_cg_cdp()
    ! eighs runs through the plugin here, but stays out of the recorded call sequence: the GPU replay applies the recorded
    ! coefficients to its own vectors, and a Lanczos recurrence with converged Ritz pairs amplifies one-ulp differences
    ! between the two states by ten orders of magnitude -- it would test that amplification, not the engine.  The pool
    ! is emptied right after (recorded), so the replay and the recording agree on it again.
    call trace_pause(.true.)
    call check_eighs_rdp()
    call check_svds_rdp()
    call trace_pause(.false.)
    call lk_gpu_release_all()
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

    !> SURVEY.md Appendix A item 4: the reference's own arnoldi on its own dense_vector, n = 1000, m = 8,
    !> d_i = 1 + (i-1)/n, x0_i = sin(i)/||.||, built against stand-in stdlib modules.  The survey recorded three H
    !> entries of that run; tests/golden/survey_reference_run_n1000_m8.npz keeps them and the oracle reproduces them to
    !> 17 digits.  This reproduces the RUN (recipe committed here); it remains a stand-in build, so it pins nothing formally.
    subroutine check_survey_values()
        integer, parameter :: ns = 1000, ms = 8
        type(diag_linop_ref) :: A
        type(dense_vector_rdp), allocatable :: X(:)
        real(dp) :: H(ms + 1, ms), x0(ns)
        integer :: i, info
        allocate (A%d(ns))
        do i = 1, ns
            A%d(i) = 1.0_dp + real(i - 1, dp)/real(ns, dp)
            x0(i) = sin(real(i, dp))
        end do
        x0 = x0/sqrt(sum(x0**2))
        allocate (X(ms + 1)); do i = 1, ms + 1; X(i)%n = ns; call X(i)%zero(); end do
        X(1) = dense_vector(x0)
        H = 0.0_dp; call arnoldi(A, X, H, info)
        call report('survey run: H(1,1)   vs recorded             ', abs(H(1, 1) - 1.4991929804973552_dp), 4.0e-15_dp)
        call report('survey run: H(2,1)   vs recorded             ', abs(H(2, 1) - 0.28878608972273856_dp), 4.0e-15_dp)
        call report('survey run: H(m+1,m) vs recorded             ', abs(H(ms + 1, ms) - 0.2505741778943683_dp), 4.0e-15_dp)
    end subroutine

    !> Runs of the REFERENCE's own arnoldi / double_gram_schmidt_step / gmres on its own dense_vector, inputs and outputs
    !> written in full precision to ref_runs.txt: tests/test_plugin_check.py feeds the same inputs to the oracle
    !> (oracle/) and compares.  An execution check that the oracle's restatement follows the reference's control flow --
    !> indicative only (stand-in stdlib), never a pin.
    subroutine dump_reference_runs()
        integer, parameter :: na = 700, ma = 12, ng = 60
        integer :: u, i, j, info
        ! arnoldi, real and complex diagonal operator
        type(diag_linop_ref) :: Ad
        type(dense_vector_rdp), allocatable :: X(:)
        real(dp) :: H(ma + 1, ma), x0(na), yv(na), beta(ma)
        type(dense_vector_rdp) :: yvec
        type(dense_linop_cdp) :: Az
        type(dense_vector_cdp), allocatable :: Xz(:)
        complex(dp) :: Hz(ma + 1, ma), x0z(ng), Amat(ng, ng)
        ! gmres
        type(dense_linop_rdp) :: Ag
        type(dense_vector_rdp) :: bg, xg
        real(dp) :: Ar(ng, ng), br(ng)
        type(gmres_dp_opts) :: opts
        type(gmres_dp_metadata) :: meta
        open (newunit=u, file='ref_runs.txt', status='replace', action='write')
        allocate (Ad%d(na))
        do i = 1, na
            Ad%d(i) = 1.0_dp + real(i - 1, dp)/real(na, dp)
            x0(i) = sin(real(3*i, dp)) + 0.25_dp
            yv(i) = cos(real(7*i, dp))
        end do
        x0 = x0/sqrt(sum(x0**2))
        allocate (X(ma + 1)); do i = 1, ma + 1; X(i)%n = na; call X(i)%zero(); end do
        X(1) = dense_vector(x0)
        H = 0.0_dp; call arnoldi(Ad, X, H, info)
        write (u, '(A,3I8)') 'case arnoldi_rdp_diag', na, ma, info
        write (u, '(ES25.17)') Ad%d, x0, H
        ! double_gram_schmidt_step of an arbitrary vector against the basis just built
        yvec = dense_vector(yv)
        call double_gram_schmidt_step(yvec, X(:ma), info, if_chk_orthonormal=.false., beta=beta)
        write (u, '(A,3I8)') 'case dgs_rdp', na, ma, info
        write (u, '(ES25.17)') yv
        do j = 1, ma
            write (u, '(ES25.17)') X(j)%data
        end do
        write (u, '(ES25.17)') beta, yvec%data
        ! complex Arnoldi with a dense complex operator (gemv restated by the oracle as a column sweep)
        do j = 1, ng
            do i = 1, ng
                Amat(i, j) = cmplx(sin(real(3*i + 7*j, dp)), cos(real(5*i - 2*j, dp)), kind=dp)/real(ng, dp)
            end do
            Amat(j, j) = Amat(j, j) + cmplx(1.0_dp + real(j, dp)/real(ng, dp), 0.3_dp, kind=dp)
            x0z(j) = cmplx(cos(real(j, dp)), sin(real(2*j, dp)), kind=dp)
        end do
        x0z = x0z/sqrt(sum(abs(x0z)**2))
        Az = dense_linop(Amat)
        allocate (Xz(ma + 1)); do i = 1, ma + 1; Xz(i)%n = ng; call Xz(i)%zero(); end do
        Xz(1) = dense_vector(x0z)
        Hz = (0.0_dp, 0.0_dp); call arnoldi(Az, Xz, Hz, info)
        write (u, '(A,3I8)') 'case arnoldi_cdp_dense', ng, ma, info
        write (u, '(2ES25.17)') Amat, x0z, Hz
        ! GMRES(20), maxiter = 2, real dense operator
        call test_matrix_small(Ar, br)
        Ag = dense_linop(Ar); bg = dense_vector(br); xg%n = ng; call xg%zero()
        opts = gmres_dp_opts(kdim=10, maxiter=2)
        call gmres(Ag, bg, xg, info, rtol=1.0e-10_dp, atol=1.0e-14_dp, options=opts, meta=meta)
        write (u, '(A,4I8)') 'case gmres_rdp_dense', ng, 10, info, size(meta%res)
        write (u, '(ES25.17)') Ar, br, xg%data, meta%res
        close (u)
    end subroutine

    subroutine test_matrix_small(A, b)
        real(dp), intent(out) :: A(:, :), b(:)
        integer :: i, j, nn
        nn = size(b)
        do j = 1, nn
            do i = 1, nn
                A(i, j) = sin(real(3*i + 7*j, dp))/real(nn, dp)
            end do
            A(j, j) = A(j, j) + 1.0_dp + real(j, dp)/real(nn, dp)
            b(j) = cos(real(j, dp))
        end do
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
        ! the adjoint operator (apply_rmatvec -> the plugin's rmatvec): arnoldi(..., transpose=.true.)   arnoldi.fypp:39-47
        do i = 2, m + 1; call Xref(i)%zero(); end do
        Xref(1) = dense_vector(x0)
        Href = 0.0_dp; call arnoldi(Lref, Xref, Href, info, transpose=.true.)
        call zero_basis(Xgpu); call Xgpu(1)%upload(x0)
        Hgpu = 0.0_dp; call arnoldi(Lgpu, Xgpu, Hgpu, info, transpose=.true.)
        call report('arnoldi rdp, transpose: max |H_gpu - H_ref| ', maxval(abs(Hgpu - Href)), 1.0e-12_dp)
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

    !> The reference's conjugate gradient through the plugin.  cg allocates its work vectors with `mold=b` (no size: the
    !> plugin infers it), copies with `p = r` (CG.fypp:116: deep copy through the handle's defined assignment) and then
    !> updates r and p separately (CG.fypp:131, 163) -- the pattern a shallow handle copy would corrupt.  The reference cannot
    !> run cg on its own dense_vector (mold= leaves dense_vector%n undefined), so the check is the textbook recurrence on
    !> plain arrays in cg's own operation order: same iterates to rounding, same iteration count.
    subroutine check_cg_rdp()
        real(dp) :: M(n, n), A(n, n), rhs(n), xg(n), xr(n), r(n), p(n), Ap(n)
        real(dp) :: alpha, beta, rr_old, rr_new, tol
        type(dense_linop_gpu_rdp) :: Lgpu
        type(sym_linop_gpu_rdp) :: S
        type(dense_vector_gpu_rdp) :: bgpu, xgpu
        type(cg_dp_opts) :: opts
        type(cg_dp_metadata) :: meta
        integer :: info, it, i
        call test_matrix_rdp(M, rhs)
        A = matmul(transpose(M), M)
        do i = 1, n; A(i, i) = A(i, i) + 1.0_dp; end do
        Lgpu = dense_linop_gpu(A); S = sym_linop_gpu(Lgpu)
        opts = cg_dp_opts(maxiter=200)
        call bgpu%upload(rhs); xgpu%n = n; call xgpu%zero()
        call cg(S, bgpu, xgpu, info, rtol=1.0e-10_dp, atol=1.0e-14_dp, options=opts, meta=meta)
        call xgpu%download(xg)
        ! plain-array restatement (CG.fypp:105-170, no preconditioner)
        tol = 1.0e-14_dp + 1.0e-10_dp*sqrt(sum(rhs**2))
        xr = 0.0_dp; r = rhs; p = r; rr_old = sum(r*r); it = 0
        do i = 1, 200
            Ap = matmul(A, p)
            alpha = rr_old/sum(p*Ap)
            xr = xr + alpha*p
            r = r - alpha*Ap
            rr_new = sum(r*r)
            it = i
            if (sqrt(rr_new) < tol) exit
            beta = rr_new/rr_old
            p = r + beta*p
            rr_old = rr_new
        end do
        call report('cg rdp: max |x_gpu - x_plain_arrays|         ', maxval(abs(xg - xr)), 1.0e-10_dp)
        call report('cg rdp: residual |A x - b|                   ', maxval(abs(matmul(A, xg) - rhs)), 1.0e-8_dp)
        call report('cg rdp: iterations differ from the restatement', real(abs(meta%n_iter - it), dp), 0.5_dp)
    end subroutine

    !> The reference's symmetric eigensolver (Lanczos + eigh of the tridiagonal matrix each step, eighs.fypp) through the
    !> plugin: its work basis is `allocate(Xwrk(kdim+1), mold=X(1))` (no size: inferred), Lanczos issues single dots and
    !> axpbys (lanczos.fypp:57-60) before the full re-orthogonalisation.  Checked against the eigenvalues of the whole matrix.
    subroutine check_eighs_rdp()
        integer, parameter :: nev = 3
        real(dp) :: M(n, n), A(n, n), rhs(n), lam_all(n), Acopy(n, n), xv(n)
        real(dp), allocatable :: lambda(:), res(:)
        type(dense_linop_gpu_rdp) :: Lgpu
        type(sym_linop_gpu_rdp) :: S
        type(dense_vector_gpu_rdp), allocatable :: X(:)
        type(dense_vector_gpu_rdp) :: x0
        integer :: info, i
        call test_matrix_rdp(M, rhs)
        A = matmul(transpose(M), M)
        do i = 1, n; A(i, i) = A(i, i) + 1.0_dp + 3.0_dp*real(i, dp)/real(n, dp); end do
        do i = 1, nev; A(i, i) = A(i, i) + 40.0_dp + 10.0_dp*real(i, dp); end do    ! three well separated leading eigenvalues
        Acopy = A; call eigh(Acopy, lam_all)
        Lgpu = dense_linop_gpu(A); S = sym_linop_gpu(Lgpu)
        allocate (X(nev)); X%n = n
        call x0%upload(rhs)
        call eighs(S, X, lambda, res, info, x0=x0, kdim=24, tolerance=1.0e-6_dp)   ! (few steps: the GPU replay of this call
        ! sequence applies these coefficients to ITS vectors, and a long Lanczos recurrence amplifies one-ulp differences)
        call report('eighs rdp: max |lambda - eig(A)| (3 largest)  ', maxval(abs(lambda(:nev) - lam_all(n:n - nev + 1:-1))), 1.0e-6_dp)
        call X(1)%download(xv)
        call report('eighs rdp: |A x - lambda x| of the leading pair', maxval(abs(matmul(A, xv) - lambda(1)*xv)), 1.0e-5_dp)
    end subroutine

    subroutine trace_pause(on)
        logical, intent(in) :: on
        interface
            function c_setenv(name, val, overwrite) bind(C, name="setenv") result(rc)
                import :: c_char, c_int
                character(kind=c_char), intent(in) :: name(*), val(*)
                integer(c_int), value :: overwrite
                integer(c_int) :: rc
            end function
        end interface
        integer(c_int) :: rc
        rc = c_setenv("LK_MOCK_TRACE_PAUSE"//c_null_char, merge("1", "0", on)//c_null_char, 1_c_int)
    end subroutine

    !> The reference's svds (Golub-Kahan bidiagonalisation: matvec AND rmatvec through the plugin, two work bases allocated with
    !> `mold=`, svd of the bidiagonal matrix each step) against the singular values of the whole matrix.
    subroutine check_svds_rdp()
        integer, parameter :: nsv = 3
        real(dp) :: A(n, n), rhs(n), sall(n), Acopy(n, n), uv(n), vv(n)
        real(dp), allocatable :: sig(:), res(:)
        type(dense_linop_gpu_rdp) :: Lgpu
        type(dense_vector_gpu_rdp), allocatable :: U(:), V(:)
        type(dense_vector_gpu_rdp) :: u0
        integer :: info, i
        call test_matrix_rdp(A, rhs)
        do i = 1, nsv; A(i, i) = A(i, i) + 20.0_dp + 10.0_dp*real(i, dp); end do     ! three well separated leading singular values
        Acopy = A; call svd(Acopy, sall)
        Lgpu = dense_linop_gpu(A)
        allocate (U(nsv)); U%n = n
        allocate (V(nsv)); V%n = n
        call u0%upload(rhs)
        call svds(Lgpu, U, sig, V, res, info, u0=u0, kdim=30, tolerance=1.0e-8_dp)
        call report('svds rdp: max |sigma - svd(A)| (3 largest)    ', maxval(abs(sig(:nsv) - sall(:nsv))), 1.0e-8_dp)
        call U(1)%download(uv); call V(1)%download(vv)
        call report('svds rdp: |A v - sigma u| of the leading triplet', maxval(abs(matmul(A, vv) - sig(1)*uv)), 1.0e-6_dp)
    end subroutine

    subroutine check_cg_cdp()
        real(dp) :: Mr(n, n), rr(n)
        complex(dp) :: M(n, n), A(n, n), rhs(n), xg(n)
        type(dense_linop_gpu_cdp) :: Lgpu
        type(hermitian_linop_gpu_cdp) :: H
        type(dense_vector_gpu_cdp) :: bgpu, xgpu
        type(cg_dp_opts) :: opts
        integer :: info, i
        call test_matrix_rdp(Mr, rr)
        M = cmplx(Mr, 0.3_dp*transpose(Mr), kind=dp); rhs = cmplx(rr, rr(n:1:-1), kind=dp)
        A = matmul(conjg(transpose(M)), M)
        do i = 1, n; A(i, i) = A(i, i) + (1.0_dp, 0.0_dp); end do
        Lgpu = dense_linop_gpu(A); H = hermitian_linop_gpu(Lgpu)
        opts = cg_dp_opts(maxiter=200)
        call bgpu%upload(rhs); xgpu%n = n; call xgpu%zero()
        call cg(H, bgpu, xgpu, info, rtol=1.0e-10_dp, atol=1.0e-14_dp, options=opts)
        call xgpu%download(xg)
        call report('cg cdp: residual |A x - b|                   ', maxval(abs(matmul(A, xg) - rhs)), 1.0e-8_dp)
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
