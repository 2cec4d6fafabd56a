!> tools/plugin_check/stale_copy.f90 -- the one idiom the plugin cannot serve, and must REFUSE loudly (tools/check_plugin.sh):
!>     allocate(b, source=dense_vector_gpu(xb))
!> `b` is a bit copy of the constructor's temporary; the next temporary created at the same address re-acquires that pool
!> column (owner tags are addresses) and overwrites it.  Without the pool's generation counter `b` silently read the new
!> temporary's data; with it, the first use of `b` stops with "stale bit copy".  Expected: exit status /= 0 and that message.
program stale_copy
    use LightKrylov_Constants, only: dp
    use LightKrylov_Logger, only: logger_setup
    use LightKrylov_AbstractVectors
    use lightkrylov_gpu
    implicit none
    integer, parameter :: n = 64
    real(dp) :: xb(n), xc(n), nb
    class(abstract_vector_rdp), allocatable :: b, c
    integer :: i
    call logger_setup(log_level=40, log_stdout=.true.)     ! errors only, on stdout: the refusal message is what the script looks for
    call lk_gpu_init(0)
    do i = 1, n
        xb(i) = 1.0_dp; xc(i) = 3.0_dp
    end do
    call make_from(b, xb)
    nb = b%norm()
    print '(a,f8.3)', 'stale_copy: |b| right after the sourced allocation = ', nb
    call make_from(c, xc)      ! same frame, same temporary address: the constructor's result re-acquires b's column
    nb = b%norm()          ! must not return |c|: the plugin stops here if b's column was handed out again
    if (abs(nb - 8.0_dp) < 1e-12_dp) then
        print '(a)', 'stale_copy: b kept its own data (the temporaries did not share an address): nothing to refuse'
        stop 0
    end if
    print '(a,f8.3)', 'stale_copy: SILENT WRONG ANSWER |b| = ', nb
    stop 3
contains
    subroutine make_from(v, x)
        class(abstract_vector_rdp), allocatable, intent(out) :: v
        real(dp), intent(in) :: x(:)
        allocate(v, source=dense_vector_gpu(x))
    end subroutine
end program
