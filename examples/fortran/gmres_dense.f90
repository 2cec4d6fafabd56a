!> A complete user program: LightKrylov's own `gmres` and `arnoldi`, unchanged, on vectors and an operator that live on the
!> MI355X through the plugin `fortran/dense_vector_gpu.f90`.
!>
!> Build inside a LightKrylov checkout (after `fpm build` of LightKrylov + fortran-lang/stdlib), e.g.
!>   amdflang -I<lightkrylov-mods> <repo>/fortran/lk_hip_iso_c.f90 <repo>/fortran/dense_vector_gpu.f90 gmres_dense.f90 \
!>            -L<lightkrylov-lib> -lLightKrylov -lstdlib -L<repo>/lightkrylov_amd -llightkrylov_hip -L/opt/rocm/lib -lamdhip64
!> In this repository it is compiled, linked and (against the host mock of the C ABI) run by tools/check_plugin.sh.
program gmres_dense
    use LightKrylov_Constants, only: dp
    use LightKrylov_Logger, only: logger_setup
    use LightKrylov_AbstractVectors
    use LightKrylov_AbstractLinops
    use LightKrylov_BaseKrylov, only: arnoldi
    use LightKrylov_IterativeSolvers, only: gmres, gmres_dp_opts
    use lightkrylov_gpu
    implicit none
    integer, parameter :: n = 200, kdim = 20
    real(dp) :: A_host(n, n), b_host(n), x_host(n), H(kdim + 1, kdim)
    type(dense_linop_gpu_rdp) :: A
    type(dense_vector_gpu_rdp) :: b, x
    type(dense_vector_gpu_rdp), allocatable :: V(:)
    type(gmres_dp_opts) :: opts
    integer :: i, j, info

    call logger_setup(log_level=100, log_stdout=.false.)
    do j = 1, n
        do i = 1, n
            A_host(i, j) = sin(real(3*i + 7*j, dp))/real(n, dp)
        end do
        A_host(j, j) = A_host(j, j) + 1.0_dp + real(j, dp)/real(n, dp)
        b_host(j) = cos(real(j, dp))
    end do

    call lk_gpu_init(device=0)                       ! one context = one GPU; switches the engine's lazy mode on
    A = dense_linop_gpu(A_host)                      ! replaces dense_linop(A_host)
    call b%upload(b_host)                            ! replaces b = dense_vector(b_host)
    x%n = n; call x%zero()

    opts = gmres_dp_opts(kdim=kdim, maxiter=10)
    call gmres(A, b, x, info, rtol=1.0e-12_dp, atol=1.0e-14_dp, options=opts)      ! LightKrylov's gmres, unmodified
    call x%download(x_host)
    print '(A,I0,A,ES10.2)', 'gmres: info = ', info, '   max |A x - b| = ', maxval(abs(matmul(A_host, x_host) - b_host))

    allocate (V(kdim + 1), source=b); call zero_basis(V)                            ! the reference's own idiom for a Krylov basis
    call V(1)%upload(b_host/sqrt(sum(b_host**2)))
    H = 0.0_dp
    call arnoldi(A, V, H, info)                                                     ! LightKrylov's arnoldi, unmodified
    print '(A,I0,A,ES22.15)', 'arnoldi: info = ', info, '   H(1,1) = ', H(1, 1)

    call lk_gpu_release_all(); call lk_gpu_finalize()
    if (maxval(abs(matmul(A_host, x_host) - b_host)) > 1.0e-9_dp) error stop 'gmres_dense: residual too large'
end program gmres_dense
