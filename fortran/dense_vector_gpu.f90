!> LightKrylov plugin: `dense_vector_gpu_{rdp,cdp}` extend LightKrylov's `abstract_vector_{rdp,cdp}`
!> (src/AbstractTypes/AbstractVectors.fypp:295-381) and `linop_gpu_{rdp,cdp}` / `dense_linop_gpu_{rdp,cdp}` extend
!> `abstract_linop_{rdp,cdp}` (src/AbstractTypes/AbstractLinops.fypp:58-87, 265-271) on top of the C ABI in
!> include/lightkrylov_hip.h, so that `arnoldi`, `gmres`, `eigs`, ... of an UNCHANGED LightKrylov run their O(n)
!> work on the MI355X.
!>
!> Build: after LightKrylov's own modules (LightKrylov + fortran-lang/stdlib; see INTEGRATION.md).  In this
!> repository tools/check_plugin.sh compiles and links it against the reference's real module interfaces and
!> executes the object-semantics logic below on the host (build container only).
!>
!> Object semantics.  LightKrylov creates vectors by sourced allocation (`allocate(V(kdim+1), source=b)`,
!> gmres.fypp:110-113; IterativeSolvers.fypp:1032) -- a BIT COPY with no user hook --, by intrinsic polymorphic
!> assignment (`wrk = V(k)`, gmres.fypp:155; `p = r`, CG.fypp:116) and by passing to `intent(out)` dummies
!> (`matvec`'s vec_out, `copy`'s out), and it never frees a vector explicitly.  Hence:
!>   * device storage is a COLUMN of a pool slab (lk_pool_* in the header), registered to an OWNER TAG = the address
!>     of the handle component inside the Fortran object.  The type has no `final`; nothing is freed behind the
!>     object's back and bit copies cannot double free.
!>   * the handle is default-initialised (unbound).  An `intent(out)` dummy is reset by the compiler on entry; its
!>     first write re-acquires by tag and gets the SAME column back, so `matvec(V(k), V(k+1))` stays in place.
!>   * a mutating call through an object whose address differs from the handle's `owner` has found a bit copy: it
!>     takes a column of its own first (copying the contents only if the operation reads them).
!>   * intrinsic assignment invokes the handle's defined `assignment(=)`: a DEEP copy, so `p = r` followed by
!>     writes to `r` leaves `p` alone (the CG update pattern).
!>   * an object that appears at the address of a dead one inherits its column: temporaries such as
!>     `linear_combination`'s `y` (allocated twice per Gram-Schmidt pass, AbstractVectors.fypp:595-598) cost no new
!>     device memory per call.  `lk_gpu_release_all()` returns everything after a solver call.
!>   Limit: a sourced copy shares its source's column until it is first written through.  Every sourced allocation
!>   in LightKrylov's Krylov layer and solvers is followed by `zero()` / `zero_basis()` or an overwrite (gmres.fypp:
!>   110-115, IterativeSolvers.fypp:1032-1034, AbstractVectors.fypp:595-598, 628-630, qr.fypp:186); user code that
!>   needs an independent copy should assign (`y = x`), not source-allocate.  The limit is ENFORCED, not silent: every
!>   pool column carries a generation counter that the pool increments each time it hands the column out, and a handle
!>   remembers the generation it was bound at.  If the source of a sourced copy dies and another object is created at its
!>   address (`allocate(b, source=dense_vector_gpu(x))`: the constructor's result is such a temporary), the copy's generation
!>   no longer matches and every use of it stops with an error instead of reading the new occupant's data.
!> Lazy batching.  Columns are handed out in the order objects are first written -- V(1), V(2), ... for
!> `allocate(V(kdim+1), source=b); call zero_basis(V)` -- and the engine runs in "lazy" mode (lk_lazy_stats,
!> lk_lazy_fusion_stats): the k calls `X(i)%dot(y)` of `innerprod` cost one panel sweep; `linear_combination`'s
!> temporary stays virtual (never written) and `y%sub(proj)` + the next `y%norm()` + the next k dots cost ONE sweep:
!> one pass over the basis per Gram-Schmidt pass.  The fully fused three-sweep DGS is reached through `gpu_arnoldi_rdp`.
module lightkrylov_gpu
    use, intrinsic :: iso_c_binding
    use, intrinsic :: iso_fortran_env, only: error_unit
    use lightkrylov_hip_c
    use LightKrylov_Constants, only: dp
    use LightKrylov_Logger, only: stop_error, type_error
    use LightKrylov_AbstractVectors, only: abstract_vector_rdp, abstract_vector_cdp
    use LightKrylov_AbstractLinops, only: abstract_linop_rdp, abstract_linop_cdp, abstract_sym_linop_rdp, &
                                          abstract_hermitian_linop_cdp
    implicit none
    private
    character(len=*), parameter :: this_module = 'LK_GPU'

    public :: dense_vector_gpu_rdp, dense_vector_gpu_cdp, dense_vector_gpu
    public :: linop_gpu_rdp, linop_gpu_cdp, dense_linop_gpu_rdp, dense_linop_gpu_cdp
    public :: dense_linop_gpu, diag_linop_gpu, diag_linspace_linop_gpu, laplacian2d_linop_gpu, ginzburg_landau_linop_gpu
    public :: csr_linop_gpu
    public :: sym_linop_gpu_rdp, hermitian_linop_gpu_cdp, sym_linop_gpu, hermitian_linop_gpu
    public :: lk_gpu_init, lk_gpu_finalize, lk_gpu_release_all, lk_gpu_context, lk_gpu_pool_stats
    public :: lk_gpu_set_partition, lk_gpu_comm_unique_id, lk_gpu_comm_init
    public :: gpu_arnoldi_rdp, gpu_arnoldi_cdp
    public :: gpu_arnoldi_segments_rdp, gpu_arnoldi_segments_cdp, lk_gpu_progress
    public :: gpu_lanczos_rdp, gpu_lanczos_cdp, gpu_bidiag_rdp, gpu_bidiag_cdp, gpu_qr_rdp, gpu_qr_cdp

    type(c_ptr), save :: ctx = c_null_ptr
    integer(c_int64_t), save :: part_row0 = 0          ! first global row of this rank's block (lk_gpu_set_partition)
    integer, save :: last_n = -1                       ! size of the vector bound most recently (resolve_size)

    !> Device storage of one vector: a column of a pool slab.  Defined assignment = deep copy.
    type :: gpu_handle
        type(c_ptr) :: buf = c_null_ptr                !! slab (panel) handle
        integer(c_int) :: col = -1                     !! this vector's column in the slab
        integer(c_int) :: dtype = -1
        integer(c_int64_t) :: n = -1
        integer(c_intptr_t) :: owner = 0               !! address of the handle this column is registered to
        integer(c_int64_t) :: gen = 0                  !! the column's generation when this handle was bound (lk_pool_column_info)
    contains
        procedure, private :: handle_assign
        generic :: assignment(=) => handle_assign
    end type

    type, extends(abstract_vector_rdp) :: dense_vector_gpu_rdp
        integer :: n = -1                   !! number of (local) rows; set by the user like dense_vector%n.  Unset (a
                                            !! `mold=` allocation, an intent(out) dummy) it is inferred: see resolve_size
        type(gpu_handle) :: h
    contains
        procedure, pass(self) :: zero => gpu_zero
        procedure, pass(self) :: rand => gpu_rand
        procedure, pass(self) :: scal => gpu_scal
        procedure, pass(self) :: axpby => gpu_axpby
        procedure, pass(self) :: dot => gpu_dot
        procedure, pass(self) :: get_size => gpu_get_size
        procedure, pass(self) :: upload => gpu_upload
        procedure, pass(self) :: download => gpu_download
        procedure, pass(self) :: device_ptr_in => gpu_ptr_in          !! for a user's own kernels: see gpu_ptr_in / gpu_ptr_out
        procedure, pass(self) :: device_ptr_out => gpu_ptr_out
    end type

    !> complex(dp) kind: same layout, interleaved (re, im) on the device (LK_C128)
    type, extends(abstract_vector_cdp) :: dense_vector_gpu_cdp
        integer :: n = -1
        type(gpu_handle) :: h
    contains
        procedure, pass(self) :: zero => gpuz_zero
        procedure, pass(self) :: rand => gpuz_rand
        procedure, pass(self) :: scal => gpuz_scal
        procedure, pass(self) :: axpby => gpuz_axpby
        procedure, pass(self) :: dot => gpuz_dot
        procedure, pass(self) :: get_size => gpuz_get_size
        procedure, pass(self) :: upload => gpuz_upload
        procedure, pass(self) :: download => gpuz_download
        procedure, pass(self) :: device_ptr_in => gpuz_ptr_in
        procedure, pass(self) :: device_ptr_out => gpuz_ptr_out
    end type

    !> Any engine operator (lk_linop_*) behind LightKrylov's abstract_linop: matvec / rmatvec = lk_linop_apply N / H.
    type, extends(abstract_linop_rdp) :: linop_gpu_rdp
        type(c_ptr) :: op = c_null_ptr
    contains
        procedure, pass(self) :: matvec => gpu_matvec_rdp
        procedure, pass(self) :: rmatvec => gpu_rmatvec_rdp
    end type
    type, extends(abstract_linop_cdp) :: linop_gpu_cdp
        type(c_ptr) :: op = c_null_ptr
    contains
        procedure, pass(self) :: matvec => gpu_matvec_cdp
        procedure, pass(self) :: rmatvec => gpu_rmatvec_cdp
    end type
    !> the same engine operator behind LightKrylov's SYMMETRIC / HERMITIAN operator types (AbstractLinops.fypp:204-256: only
    !> `matvec` is deferred), which `cg` and `eighs` require: S = sym_linop_gpu(L), H = hermitian_linop_gpu(L) share L's handle
    type, extends(abstract_sym_linop_rdp) :: sym_linop_gpu_rdp
        type(c_ptr) :: op = c_null_ptr
    contains
        procedure, pass(self) :: matvec => gpu_sym_matvec_rdp
    end type
    type, extends(abstract_hermitian_linop_cdp) :: hermitian_linop_gpu_cdp
        type(c_ptr) :: op = c_null_ptr
    contains
        procedure, pass(self) :: matvec => gpu_herm_matvec_cdp
    end type
    !> dense_linop on the device (AbstractLinops.fypp:265-271, 608-660), both double-precision kinds
    type, extends(linop_gpu_rdp) :: dense_linop_gpu_rdp
    end type
    type, extends(linop_gpu_cdp) :: dense_linop_gpu_cdp
    end type

    !> dense_vector_gpu(x): the counterpart of the reference's dense_vector(x) constructor (AbstractVectors.fypp:469-474)
    interface dense_vector_gpu
        module procedure dense_vector_gpu_from_rdp, dense_vector_gpu_from_cdp
    end interface
    !> dense_linop_gpu(A): the whole matrix on one rank.  dense_linop_gpu(A_rows, row_starts): ROW-SHARDED -- this rank passes its
    !> n_local x n_global row block and the 0-based offsets row_starts(0:nranks) of every rank's block (lk_gpu_comm_init first):
    !> matvec all-gathers x, rmatvec sums the ranks' products of their blocks' conjugate transposes (lightkrylov_hip.h).
    interface dense_linop_gpu
        module procedure dense_linop_gpu_from_rdp, dense_linop_gpu_from_cdp, dense_linop_gpu_rows_rdp, dense_linop_gpu_rows_cdp
    end interface
    interface diag_linop_gpu
        module procedure diag_linop_gpu_from_rdp, diag_linop_gpu_from_cdp
    end interface
    !> a sparse operator in CSR with Fortran's 1-based `rowptr(n+1)` / `colind(nnz)`; matvec and rmatvec on the device
    interface csr_linop_gpu
        module procedure csr_linop_gpu_from_rdp, csr_linop_gpu_from_cdp
    end interface

    !> progress function of gpu_arnoldi_segments_*: called with the first and last step whose columns of H have just become final;
    !> a non-zero return value asks the engine to stop (at most 24 further steps have been enqueued)
    abstract interface
        function lk_gpu_progress(user, kfirst, klast) bind(C) result(stop)
            import :: c_ptr, c_int
            type(c_ptr), value :: user
            integer(c_int), value :: kfirst, klast
            integer(c_int) :: stop
        end function
    end interface

contains

    ! ---- context -----------------------------------------------------------------------------------------
    subroutine lk_gpu_init(device)
        integer, intent(in) :: device
        call chk(lk_init(int(device, c_int), c_null_ptr, ctx), 'lk_gpu_init')
        call chk(lk_set_tuning(ctx, 'lazy'//c_null_char, 1_c_int), 'lk_gpu_init')
        part_row0 = 0
    end subroutine

    function lk_gpu_context() result(c)
        type(c_ptr) :: c
        c = ctx
    end function

    !> returns every pool column (all vectors become unbound; their next write re-acquires)
    subroutine lk_gpu_release_all()
        call chk(lk_pool_release_all(ctx), 'lk_gpu_release_all')
    end subroutine

    subroutine lk_gpu_finalize()
        integer(c_int) :: rc
        rc = lk_finalize(ctx); ctx = c_null_ptr
    end subroutine

    !> out4 = slabs, columns ever carved, columns currently registered, acquisitions served by re-use
    subroutine lk_gpu_pool_stats(out4)
        integer(c_int64_t), intent(out) :: out4(4)
        call chk(lk_pool_stats(ctx, out4), 'lk_gpu_pool_stats')
    end subroutine

    !> this rank owns global rows [row0, row0 + n_local) of n_global (row-sharded run, one process per GPU)
    subroutine lk_gpu_set_partition(row0, n_global)
        integer(c_int64_t), intent(in) :: row0, n_global
        call chk(lk_set_partition(ctx, row0, n_global), 'lk_gpu_set_partition')
        part_row0 = row0
    end subroutine

    !> rank 0: 128 opaque bytes to broadcast (MPI_Bcast) before every rank calls lk_gpu_comm_init
    subroutine lk_gpu_comm_unique_id(id)
        character(kind=c_char), intent(out) :: id(128)
        call chk(lk_comm_get_unique_id(id), 'lk_gpu_comm_unique_id')
    end subroutine

    !> collective: native RCCL sum all-reduce for every dot / norm / Gram-Schmidt coefficient from now on
    subroutine lk_gpu_comm_init(nranks, rank, id)
        integer, intent(in) :: nranks, rank
        character(kind=c_char), intent(in) :: id(128)
        call chk(lk_comm_init_rank(ctx, int(nranks, c_int), int(rank, c_int), id), 'lk_gpu_comm_init')
    end subroutine

    !> the offsets of a row-sharded operator: exactly nranks + 1 of them, from 0 to the global size (the library reads
    !> row_starts(rank + 1) and row_starts(nranks): a shorter array would be read out of bounds)
    subroutine check_row_starts(row_starts, n_global, procedure)
        integer(c_int64_t), intent(in) :: row_starts(0:)
        integer, intent(in) :: n_global
        character(len=*), intent(in) :: procedure
        integer(c_int) :: nranks, rank
        ! the rank count comes from the LIBRARY (lk_comm_info), not from a shadow of this module: a context whose collectives were
        ! installed through the iso_c bindings (lk_set_allreduce / lk_set_allgather) has ranks this module never saw
        call chk(lk_comm_info(ctx, nranks, rank), procedure)
        if (size(row_starts) /= nranks + 1) call stop_error( &
            'row_starts needs nranks + 1 entries (0-based offsets of every rank''s row block; install the communicator first)', this_module, procedure)
        if (row_starts(0) /= 0 .or. row_starts(nranks) /= int(n_global, c_int64_t)) call stop_error( &
            'row_starts must run from 0 to the number of columns of the row block', this_module, procedure)
    end subroutine

    subroutine chk(rc, procedure)
        integer(c_int), intent(in) :: rc
        character(len=*), intent(in) :: procedure
        ! non-zero status => LightKrylov's fatal path (Logger.f90:290-298): log_error + STOP 1
        if (rc /= LK_OK) call stop_error(lk_error_message(), this_module, procedure)
    end subroutine

    !> Size of a vector that does not know it yet.  `allocate(r, mold=b)` (CG.fypp:113-121, eighs.fypp:60, svd_solvers: the
    !> work vectors of cg / eighs / svds) gives an object of the right TYPE but no size, and the first call on it is
    !> `zero()`: the plugin then assumes the size of the vector bound most recently (one problem size per program is the
    !> rule; with several sizes in play allocate with `source=` or set `%n`).  Operators and `axpby` / `copy` take the size
    !> from their input instead (apply_rdp, gpu_axpby), as dense_axpby does (AbstractVectors.fypp:521-524).
    subroutine resolve_size(n)
        integer, intent(inout) :: n
        if (n < 0) n = last_n
    end subroutine

    ! ---- handle ----------------------------------------------------------------------------------------------
    !> .true. when `h` is registered to the object it sits in (not a bit copy, not stale) with this shape
    logical function handle_is_own(h, dtype, n) result(own)
        type(gpu_handle), intent(in), target :: h
        integer(c_int), intent(in) :: dtype
        integer(c_int64_t), intent(in) :: n
        integer(c_intptr_t) :: tag, reg
        integer(c_int64_t) :: gen
        own = .false.
        if (.not. c_associated(h%buf)) return
        tag = transfer(c_loc(h), tag)
        if (h%owner /= tag .or. h%dtype /= dtype .or. h%n /= n) return
        call chk(lk_pool_column_info(ctx, h%buf, h%col, reg, gen), 'handle_is_own')
        own = (reg == tag .and. gen == h%gen)
    end function

    !> .true. when `h` points at a registered pool column (own or shared) that has not been handed out again since `h` was
    !> bound: safe to read.  A bit copy whose source object died and was replaced at the same address (the source of
    !> `allocate(b, source=dense_vector_gpu(x))` is such a temporary) fails the generation test: the caller stops with an
    !> error instead of reading the new occupant's data.
    logical function handle_is_readable(h) result(ok)
        type(gpu_handle), intent(in) :: h
        integer(c_intptr_t) :: reg
        integer(c_int64_t) :: gen
        ok = .false.
        if (.not. c_associated(h%buf)) return
        call chk(lk_pool_column_info(ctx, h%buf, h%col, reg, gen), 'handle_is_readable')
        ok = (reg /= 0 .and. gen == h%gen)
    end function

    !> .true. when `h` points at a pool column that has been handed out AGAIN since `h` was bound (see handle_is_readable)
    logical function handle_is_stale_copy(h) result(stale)
        type(gpu_handle), intent(in) :: h
        integer(c_intptr_t) :: reg
        integer(c_int64_t) :: gen
        stale = .false.
        if (.not. c_associated(h%buf)) return
        call chk(lk_pool_column_info(ctx, h%buf, h%col, reg, gen), 'handle_is_stale_copy')
        stale = (reg /= 0 .and. gen /= h%gen)
    end function

    !> Stop unless `h` can be read: "stale bit copy" when its column has been handed out again since (see handle_is_readable),
    !> "<what> holds no data" when it was never bound or the pool was released.
    subroutine require_readable(h, what, procedure)
        type(gpu_handle), intent(in) :: h
        character(len=*), intent(in) :: what, procedure
        character(len=*), parameter :: stale = ' is a stale bit copy: the object it was source-allocated from has been '// &
            're-initialised or replaced; make independent copies by assignment (y = x), not by allocate(y, source=x)'
        if (handle_is_readable(h)) return
        if (handle_is_stale_copy(h)) then
            write (error_unit, '(a)') this_module//' % '//procedure//': '//what//stale    ! stop_error logs the text only at debug level
            call stop_error(what//stale, this_module, procedure)
        end if
        call stop_error(what//' holds no data', this_module, procedure)
    end subroutine

    !> Make `h` own a column of shape (dtype, n); keep=.true. preserves what it could read before.
    subroutine handle_bind(h, dtype, n, keep)
        type(gpu_handle), intent(inout), target :: h
        integer(c_int), intent(in) :: dtype
        integer(c_int64_t), intent(in) :: n
        logical, intent(in) :: keep
        type(c_ptr) :: fresh, old_buf
        integer(c_int) :: fresh_col, old_col
        integer(c_intptr_t) :: tag, reg
        logical :: copy_old
        if (n <= 0) call stop_error('vector size not set (set %n or upload first)', this_module, 'bind')
        if (handle_is_own(h, dtype, n)) then
            last_n = int(n)
            return
        end if
        copy_old = .false.
        if (keep .and. h%dtype == dtype .and. h%n == n) then
            copy_old = handle_is_readable(h)
            if (.not. copy_old .and. handle_is_stale_copy(h)) call stop_error( &
                'stale bit copy: the object this vector was source-allocated from has been re-initialised; assign (y = x) instead', &
                this_module, 'bind')
        end if
        old_buf = h%buf; old_col = h%col
        tag = transfer(c_loc(h), tag)
        call chk(lk_pool_acquire(ctx, dtype, n, tag, fresh, fresh_col), 'bind')
        if (copy_old) then
            if (.not. (c_associated(fresh, old_buf) .and. fresh_col == old_col)) &
                call chk(lk_vec_copy(fresh, fresh_col, old_buf, old_col), 'bind')
        end if
        h%buf = fresh; h%col = fresh_col; h%dtype = dtype; h%n = n; h%owner = tag
        call chk(lk_pool_column_info(ctx, fresh, fresh_col, reg, h%gen), 'bind')
        last_n = int(n)
    end subroutine

    !> defined assignment of the handle component = DEEP COPY (intrinsic assignment `wrk = V(k)`, `p = r`)
    subroutine handle_assign(lhs, rhs)
        class(gpu_handle), intent(inout), target :: lhs
        class(gpu_handle), intent(in) :: rhs
        if (.not. c_associated(rhs%buf)) then          ! unbound source: unbound copy
            lhs%buf = c_null_ptr; lhs%col = -1; lhs%dtype = rhs%dtype; lhs%n = rhs%n; lhs%owner = 0; lhs%gen = 0
            return
        end if
        if (.not. handle_is_readable(rhs)) then        ! stale source (pool released): nothing to copy
            lhs%buf = c_null_ptr; lhs%col = -1; lhs%dtype = rhs%dtype; lhs%n = rhs%n; lhs%owner = 0; lhs%gen = 0
            return
        end if
        select type (lhs)
        type is (gpu_handle)
            call handle_bind(lhs, rhs%dtype, rhs%n, .false.)
        end select
        if (.not. (c_associated(lhs%buf, rhs%buf) .and. lhs%col == rhs%col)) &
            call chk(lk_vec_copy(lhs%buf, lhs%col, rhs%buf, rhs%col), 'assignment(=)')
    end subroutine

    function dense_vector_gpu_from_rdp(x) result(vec)
        real(dp), intent(in) :: x(:)
        type(dense_vector_gpu_rdp) :: vec
        call vec%upload(x)
    end function
    function dense_vector_gpu_from_cdp(x) result(vec)
        complex(dp), intent(in) :: x(:)
        type(dense_vector_gpu_cdp) :: vec
        call vec%upload(x)
    end function

    ! ---- real(dp) kind ---------------------------------------------------------------------------------------
    subroutine gpu_zero(self)
        class(dense_vector_gpu_rdp), intent(inout) :: self
        call resolve_size(self%n)
        call handle_bind(self%h, LK_F64, int(self%n, c_int64_t), .false.)
        call chk(lk_vec_zero(self%h%buf, self%h%col), 'zero')
    end subroutine

    subroutine gpu_rand(self, ifnorm)
        class(dense_vector_gpu_rdp), intent(inout) :: self
        logical, optional, intent(in) :: ifnorm
        integer(c_int) :: nrm
        integer(c_int64_t), save :: seed = 1
        nrm = 0; if (present(ifnorm)) nrm = merge(1_c_int, 0_c_int, ifnorm)
        call resolve_size(self%n)
        call handle_bind(self%h, LK_F64, int(self%n, c_int64_t), .false.)
        seed = seed + 1
        call chk(lk_vec_rand(self%h%buf, self%h%col, seed, part_row0, nrm), 'rand')
    end subroutine

    subroutine gpu_scal(self, alpha)
        class(dense_vector_gpu_rdp), intent(inout) :: self
        real(dp), intent(in) :: alpha
        call handle_bind(self%h, LK_F64, int(self%n, c_int64_t), .true.)
        call chk(lk_vec_scal(self%h%buf, self%h%col, [alpha]), 'scal')
    end subroutine

    subroutine gpu_axpby(alpha, vec, beta, self)
        real(dp), intent(in) :: alpha, beta
        class(abstract_vector_rdp), intent(in) :: vec
        class(dense_vector_gpu_rdp), intent(inout) :: self
        select type (vec)
        class is (dense_vector_gpu_rdp)
            ! no storage yet (fresh object, or an intent(out) dummy such as copy's `out`): adopt vec's size, like
            ! dense_axpby's `if (.not. allocated(self%data)) allocate(self%data(m))` (AbstractVectors.fypp:521-524)
            if (.not. c_associated(self%h%buf)) self%n = vec%n
            if (vec%n /= self%n) call stop_error("Inconsistent size between the two vectors.", this_module, 'axpby')
            call require_readable(vec%h, 'vec', 'axpby')
            call handle_bind(self%h, LK_F64, int(self%n, c_int64_t), beta /= 0.0_dp)   ! beta == 0: old contents are not read
            call chk(lk_vec_axpby([alpha], vec%h%buf, vec%h%col, [beta], self%h%buf, self%h%col), 'axpby')
        class default
            call type_error('vec', 'dense_vector_gpu_rdp', 'IN', this_module, 'axpby')
        end select
    end subroutine

    function gpu_dot(self, vec) result(alpha)
        class(dense_vector_gpu_rdp), intent(in) :: self
        class(abstract_vector_rdp), intent(in) :: vec
        real(dp) :: alpha
        real(c_double) :: res(2)
        alpha = 0.0_dp
        select type (vec)
        class is (dense_vector_gpu_rdp)
            if (vec%n /= self%n) call stop_error("Inconsistent size between the two vectors.", this_module, 'dot')
            call require_readable(self%h, 'self', 'dot')
            call require_readable(vec%h, 'vec', 'dot')
            call chk(lk_vec_dot(self%h%buf, self%h%col, vec%h%buf, vec%h%col, res), 'dot')
            alpha = res(1)
        class default
            call type_error('vec', 'dense_vector_gpu_rdp', 'IN', this_module, 'dot')
        end select
    end function

    function gpu_get_size(self) result(n)
        class(dense_vector_gpu_rdp), intent(in) :: self
        integer :: n
        n = self%n
    end function

    subroutine gpu_upload(self, x)
        class(dense_vector_gpu_rdp), intent(inout) :: self
        real(dp), intent(in), target :: x(:)
        self%n = size(x)
        call handle_bind(self%h, LK_F64, int(self%n, c_int64_t), .false.)
        call chk(lk_basis_upload(self%h%buf, self%h%col, 1_c_int, c_loc(x), int(self%n, c_int64_t)), 'upload')
    end subroutine

    !> A user's own `matvec(self, vec_in, vec_out)` written with device kernels (hipfort, OpenMP target `is_device_ptr`, a C
    !> wrapper ...): `vec_in%device_ptr_in()` is the device address of the n real(dp) values to READ, `vec_out%device_ptr_out()`
    !> the address to WRITE all n values to (vec_out is intent(out): it gets a column of its own here, previous contents are
    !> not kept).  Pending engine work on the vector is applied first (lk_vec_device_ptr); the kernel must run on the engine's
    !> stream (lk_context_info) or after lk_sync.
    function gpu_ptr_in(self) result(p)
        class(dense_vector_gpu_rdp), intent(in) :: self
        type(c_ptr) :: p
        call require_readable(self%h, 'vector', 'device_ptr_in')
        call chk(lk_vec_device_ptr(self%h%buf, self%h%col, LK_ACCESS_READ, p), 'device_ptr_in')
    end function
    function gpu_ptr_out(self) result(p)
        class(dense_vector_gpu_rdp), intent(inout) :: self
        type(c_ptr) :: p
        call resolve_size(self%n)
        call handle_bind(self%h, LK_F64, int(self%n, c_int64_t), .false.)
        call chk(lk_vec_device_ptr(self%h%buf, self%h%col, LK_ACCESS_OVERWRITE, p), 'device_ptr_out')
    end function
    function gpuz_ptr_in(self) result(p)
        class(dense_vector_gpu_cdp), intent(in) :: self
        type(c_ptr) :: p
        call require_readable(self%h, 'vector', 'device_ptr_in')
        call chk(lk_vec_device_ptr(self%h%buf, self%h%col, LK_ACCESS_READ, p), 'device_ptr_in')
    end function
    function gpuz_ptr_out(self) result(p)
        class(dense_vector_gpu_cdp), intent(inout) :: self
        type(c_ptr) :: p
        call resolve_size(self%n)
        call handle_bind(self%h, LK_C128, int(self%n, c_int64_t), .false.)
        call chk(lk_vec_device_ptr(self%h%buf, self%h%col, LK_ACCESS_OVERWRITE, p), 'device_ptr_out')
    end function

    subroutine gpu_download(self, x)
        class(dense_vector_gpu_rdp), intent(in) :: self
        real(dp), intent(out), target :: x(:)
        call require_readable(self%h, 'vector', 'download')
        call chk(lk_basis_download(self%h%buf, self%h%col, 1_c_int, c_loc(x), int(self%n, c_int64_t)), 'download')
    end subroutine

    ! ---- complex(dp) kind ------------------------------------------------------------------------------------
    subroutine gpuz_zero(self)
        class(dense_vector_gpu_cdp), intent(inout) :: self
        call resolve_size(self%n)
        call handle_bind(self%h, LK_C128, int(self%n, c_int64_t), .false.)
        call chk(lk_vec_zero(self%h%buf, self%h%col), 'zero')
    end subroutine

    subroutine gpuz_rand(self, ifnorm)
        class(dense_vector_gpu_cdp), intent(inout) :: self
        logical, optional, intent(in) :: ifnorm
        integer(c_int) :: nrm
        integer(c_int64_t), save :: seed = 1000001
        nrm = 0; if (present(ifnorm)) nrm = merge(1_c_int, 0_c_int, ifnorm)
        call resolve_size(self%n)
        call handle_bind(self%h, LK_C128, int(self%n, c_int64_t), .false.)
        seed = seed + 1
        call chk(lk_vec_rand(self%h%buf, self%h%col, seed, part_row0, nrm), 'rand')
    end subroutine

    subroutine gpuz_scal(self, alpha)
        class(dense_vector_gpu_cdp), intent(inout) :: self
        complex(dp), intent(in) :: alpha
        call handle_bind(self%h, LK_C128, int(self%n, c_int64_t), .true.)
        call chk(lk_vec_scal(self%h%buf, self%h%col, [real(alpha, dp), aimag(alpha)]), 'scal')
    end subroutine

    subroutine gpuz_axpby(alpha, vec, beta, self)
        complex(dp), intent(in) :: alpha, beta
        class(abstract_vector_cdp), intent(in) :: vec
        class(dense_vector_gpu_cdp), intent(inout) :: self
        select type (vec)
        class is (dense_vector_gpu_cdp)
            if (.not. c_associated(self%h%buf)) self%n = vec%n
            if (vec%n /= self%n) call stop_error("Inconsistent size between the two vectors.", this_module, 'axpby')
            call require_readable(vec%h, 'vec', 'axpby')
            call handle_bind(self%h, LK_C128, int(self%n, c_int64_t), beta /= (0.0_dp, 0.0_dp))
            call chk(lk_vec_axpby([real(alpha, dp), aimag(alpha)], vec%h%buf, vec%h%col, [real(beta, dp), aimag(beta)], &
                                  self%h%buf, self%h%col), 'axpby')
        class default
            call type_error('vec', 'dense_vector_gpu_cdp', 'IN', this_module, 'axpby')
        end select
    end subroutine

    function gpuz_dot(self, vec) result(alpha)
        class(dense_vector_gpu_cdp), intent(in) :: self
        class(abstract_vector_cdp), intent(in) :: vec
        complex(dp) :: alpha
        real(c_double) :: res(2)
        alpha = (0.0_dp, 0.0_dp)
        select type (vec)
        class is (dense_vector_gpu_cdp)
            if (vec%n /= self%n) call stop_error("Inconsistent size between the two vectors.", this_module, 'dot')
            call require_readable(self%h, 'self', 'dot')
            call require_readable(vec%h, 'vec', 'dot')
            call chk(lk_vec_dot(self%h%buf, self%h%col, vec%h%buf, vec%h%col, res), 'dot')   ! conj on self, like dotc
            alpha = cmplx(res(1), res(2), kind=dp)
        class default
            call type_error('vec', 'dense_vector_gpu_cdp', 'IN', this_module, 'dot')
        end select
    end function

    function gpuz_get_size(self) result(n)
        class(dense_vector_gpu_cdp), intent(in) :: self
        integer :: n
        n = self%n
    end function

    subroutine gpuz_upload(self, x)
        class(dense_vector_gpu_cdp), intent(inout) :: self
        complex(dp), intent(in), target :: x(:)
        self%n = size(x)
        call handle_bind(self%h, LK_C128, int(self%n, c_int64_t), .false.)
        call chk(lk_basis_upload(self%h%buf, self%h%col, 1_c_int, c_loc(x), int(self%n, c_int64_t)), 'upload')
    end subroutine

    subroutine gpuz_download(self, x)
        class(dense_vector_gpu_cdp), intent(in) :: self
        complex(dp), intent(out), target :: x(:)
        call require_readable(self%h, 'vector', 'download')
        call chk(lk_basis_download(self%h%buf, self%h%col, 1_c_int, c_loc(x), int(self%n, c_int64_t)), 'download')
    end subroutine

    ! ---- operators ---------------------------------------------------------------------------------------------
    subroutine apply_rdp(op, trans, vec_in, vec_out, procedure)
        type(c_ptr), intent(in) :: op
        integer(c_int), intent(in) :: trans
        class(abstract_vector_rdp), intent(in) :: vec_in
        class(abstract_vector_rdp), intent(inout) :: vec_out   ! the TBP's intent(out) dummy has already been reset
        character(len=*), intent(in) :: procedure
        select type (vec_in)
        class is (dense_vector_gpu_rdp)
            select type (vec_out)
            class is (dense_vector_gpu_rdp)
                call require_readable(vec_in%h, 'vec_in', procedure)
                vec_out%n = vec_in%n
                call handle_bind(vec_out%h, LK_F64, int(vec_out%n, c_int64_t), .false.)
                call chk(lk_linop_apply(op, trans, vec_in%h%buf, vec_in%h%col, vec_out%h%buf, vec_out%h%col), procedure)
            class default
                call type_error('vec_out', 'dense_vector_gpu_rdp', 'OUT', this_module, procedure)
            end select
        class default
            call type_error('vec_in', 'dense_vector_gpu_rdp', 'IN', this_module, procedure)
        end select
    end subroutine

    subroutine apply_cdp(op, trans, vec_in, vec_out, procedure)
        type(c_ptr), intent(in) :: op
        integer(c_int), intent(in) :: trans
        class(abstract_vector_cdp), intent(in) :: vec_in
        class(abstract_vector_cdp), intent(inout) :: vec_out
        character(len=*), intent(in) :: procedure
        select type (vec_in)
        class is (dense_vector_gpu_cdp)
            select type (vec_out)
            class is (dense_vector_gpu_cdp)
                call require_readable(vec_in%h, 'vec_in', procedure)
                vec_out%n = vec_in%n
                call handle_bind(vec_out%h, LK_C128, int(vec_out%n, c_int64_t), .false.)
                call chk(lk_linop_apply(op, trans, vec_in%h%buf, vec_in%h%col, vec_out%h%buf, vec_out%h%col), procedure)
            class default
                call type_error('vec_out', 'dense_vector_gpu_cdp', 'OUT', this_module, procedure)
            end select
        class default
            call type_error('vec_in', 'dense_vector_gpu_cdp', 'IN', this_module, procedure)
        end select
    end subroutine

    subroutine gpu_matvec_rdp(self, vec_in, vec_out)
        class(linop_gpu_rdp), intent(inout) :: self
        class(abstract_vector_rdp), intent(in) :: vec_in
        class(abstract_vector_rdp), intent(out) :: vec_out
        call apply_rdp(self%op, LK_OP_N, vec_in, vec_out, 'matvec')
    end subroutine
    subroutine gpu_rmatvec_rdp(self, vec_in, vec_out)
        class(linop_gpu_rdp), intent(inout) :: self
        class(abstract_vector_rdp), intent(in) :: vec_in
        class(abstract_vector_rdp), intent(out) :: vec_out
        call apply_rdp(self%op, LK_OP_H, vec_in, vec_out, 'rmatvec')
    end subroutine
    subroutine gpu_sym_matvec_rdp(self, vec_in, vec_out)
        class(sym_linop_gpu_rdp), intent(inout) :: self
        class(abstract_vector_rdp), intent(in) :: vec_in
        class(abstract_vector_rdp), intent(out) :: vec_out
        call apply_rdp(self%op, LK_OP_N, vec_in, vec_out, 'matvec')
    end subroutine
    subroutine gpu_herm_matvec_cdp(self, vec_in, vec_out)
        class(hermitian_linop_gpu_cdp), intent(inout) :: self
        class(abstract_vector_cdp), intent(in) :: vec_in
        class(abstract_vector_cdp), intent(out) :: vec_out
        call apply_cdp(self%op, LK_OP_N, vec_in, vec_out, 'matvec')
    end subroutine
    function sym_linop_gpu(L) result(S)
        class(linop_gpu_rdp), intent(in) :: L
        type(sym_linop_gpu_rdp) :: S
        S%op = L%op
    end function
    function hermitian_linop_gpu(L) result(H)
        class(linop_gpu_cdp), intent(in) :: L
        type(hermitian_linop_gpu_cdp) :: H
        H%op = L%op
    end function
    subroutine gpu_matvec_cdp(self, vec_in, vec_out)
        class(linop_gpu_cdp), intent(inout) :: self
        class(abstract_vector_cdp), intent(in) :: vec_in
        class(abstract_vector_cdp), intent(out) :: vec_out
        call apply_cdp(self%op, LK_OP_N, vec_in, vec_out, 'matvec')
    end subroutine
    subroutine gpu_rmatvec_cdp(self, vec_in, vec_out)
        class(linop_gpu_cdp), intent(inout) :: self
        class(abstract_vector_cdp), intent(in) :: vec_in
        class(abstract_vector_cdp), intent(out) :: vec_out
        call apply_cdp(self%op, LK_OP_H, vec_in, vec_out, 'rmatvec')
    end subroutine

    ! constructors: the operator lives in the engine; the Fortran object holds its handle
    function dense_linop_gpu_from_rdp(A) result(L)
        real(dp), intent(in), target :: A(:, :)
        type(dense_linop_gpu_rdp) :: L
        call chk(lk_linop_dense_create(ctx, LK_F64, int(size(A, 1), c_int64_t), c_loc(A), int(size(A, 1), c_int64_t), L%op), &
                 'dense_linop_gpu')
    end function
    function dense_linop_gpu_from_cdp(A) result(L)
        complex(dp), intent(in), target :: A(:, :)
        type(dense_linop_gpu_cdp) :: L
        call chk(lk_linop_dense_create(ctx, LK_C128, int(size(A, 1), c_int64_t), c_loc(A), int(size(A, 1), c_int64_t), L%op), &
                 'dense_linop_gpu')
    end function
    function dense_linop_gpu_rows_rdp(A_rows, row_starts) result(L)
        real(dp), intent(in), target :: A_rows(:, :)
        integer(c_int64_t), intent(in) :: row_starts(0:)
        type(dense_linop_gpu_rdp) :: L
        call check_row_starts(row_starts, size(A_rows, 2), 'dense_linop_gpu (row block)')
        call chk(lk_linop_dense_create_sharded(ctx, LK_F64, int(size(A_rows, 2), c_int64_t), row_starts, c_loc(A_rows), &
                                               int(max(size(A_rows, 1), 1), c_int64_t), L%op), 'dense_linop_gpu (row block)')
    end function
    function dense_linop_gpu_rows_cdp(A_rows, row_starts) result(L)
        complex(dp), intent(in), target :: A_rows(:, :)
        integer(c_int64_t), intent(in) :: row_starts(0:)
        type(dense_linop_gpu_cdp) :: L
        call check_row_starts(row_starts, size(A_rows, 2), 'dense_linop_gpu (row block)')
        call chk(lk_linop_dense_create_sharded(ctx, LK_C128, int(size(A_rows, 2), c_int64_t), row_starts, c_loc(A_rows), &
                                               int(max(size(A_rows, 1), 1), c_int64_t), L%op), 'dense_linop_gpu (row block)')
    end function
    function csr_linop_gpu_from_rdp(rowptr, colind, vals) result(L)
        integer, intent(in) :: rowptr(:), colind(:)
        real(dp), intent(in), target :: vals(:)
        type(linop_gpu_rdp) :: L
        integer(c_int64_t), allocatable, target :: rp(:)
        integer(c_int32_t), allocatable, target :: ci(:)
        rp = int(rowptr, c_int64_t) - 1_c_int64_t
        ci = int(colind, c_int32_t) - 1_c_int32_t
        call chk(lk_linop_csr_create(ctx, LK_F64, int(size(rowptr) - 1, c_int64_t), c_loc(rp), c_loc(ci), c_loc(vals), L%op), &
                 'csr_linop_gpu')
    end function
    function csr_linop_gpu_from_cdp(rowptr, colind, vals) result(L)
        integer, intent(in) :: rowptr(:), colind(:)
        complex(dp), intent(in), target :: vals(:)
        type(linop_gpu_cdp) :: L
        integer(c_int64_t), allocatable, target :: rp(:)
        integer(c_int32_t), allocatable, target :: ci(:)
        rp = int(rowptr, c_int64_t) - 1_c_int64_t
        ci = int(colind, c_int32_t) - 1_c_int32_t
        call chk(lk_linop_csr_create(ctx, LK_C128, int(size(rowptr) - 1, c_int64_t), c_loc(rp), c_loc(ci), c_loc(vals), L%op), &
                 'csr_linop_gpu')
    end function
    function diag_linop_gpu_from_rdp(d) result(L)
        real(dp), intent(in), target :: d(:)
        type(linop_gpu_rdp) :: L
        call chk(lk_linop_diag_create(ctx, LK_F64, int(size(d), c_int64_t), c_loc(d), L%op), 'diag_linop_gpu')
    end function
    function diag_linop_gpu_from_cdp(d) result(L)
        complex(dp), intent(in), target :: d(:)
        type(linop_gpu_cdp) :: L
        call chk(lk_linop_diag_create(ctx, LK_C128, int(size(d), c_int64_t), c_loc(d), L%op), 'diag_linop_gpu')
    end function
    !> d_i = d0 + dstep*(row0 + i), generated on the device (row-sharded runs pass this rank's row0)
    function diag_linspace_linop_gpu(n_local, row0, d0, dstep) result(L)
        integer(c_int64_t), intent(in) :: n_local, row0
        real(dp), intent(in) :: d0, dstep
        type(linop_gpu_rdp) :: L
        call chk(lk_linop_diag_linspace_create(ctx, n_local, row0, d0, dstep, L%op), 'diag_linspace_linop_gpu')
    end function
    !> 5-point Laplacian on an N x N grid, Dirichlet, scaled by (N+1)^2.  Row-sharded runs pass the grid lines this rank
    !> owns (j0 = first line, 0-based; nj = number of lines): one line is exchanged with each neighbouring rank per matvec.
    function laplacian2d_linop_gpu(N, j0, nj) result(L)
        integer, intent(in) :: N
        integer, optional, intent(in) :: j0, nj
        type(linop_gpu_rdp) :: L
        if (present(j0) .and. present(nj)) then
            call chk(lk_linop_lap5_create_sharded(ctx, int(N, c_int64_t), int(j0, c_int64_t), int(nj, c_int64_t), L%op), &
                     'laplacian2d_linop_gpu')
        else
            call chk(lk_linop_lap5_create(ctx, int(N, c_int64_t), L%op), 'laplacian2d_linop_gpu')
        end if
    end function
    !> fixed-step RK4 propagator of the linearised Ginzburg-Landau operator (example/ginzburg_landau/Ginzburg_Landau.f90:126-136);
    !> n = GLOBAL size; row-sharded runs pass this rank's block (row0 0-based, n_local): one point per RK4 stage goes to each
    !> neighbouring rank.
    function ginzburg_landau_linop_gpu(n, dx, tau, nsub, nu, gamma, mu_c, mu2, row0, n_local) result(L)
        integer, intent(in) :: n, nsub
        real(dp), intent(in) :: dx, tau, mu_c, mu2
        complex(dp), intent(in) :: nu, gamma
        integer, optional, intent(in) :: row0, n_local
        type(linop_gpu_cdp) :: L
        if (present(row0) .and. present(n_local)) then
            call chk(lk_linop_gl_create_sharded(ctx, int(n, c_int64_t), int(row0, c_int64_t), int(n_local, c_int64_t), dx, tau, &
                                                int(nsub, c_int), [real(nu, dp), aimag(nu)], [real(gamma, dp), aimag(gamma)], &
                                                mu_c, mu2, L%op), 'ginzburg_landau_linop_gpu')
        else
            call chk(lk_linop_gl_create(ctx, int(n, c_int64_t), dx, tau, int(nsub, c_int), [real(nu, dp), aimag(nu)], &
                                        [real(gamma, dp), aimag(gamma)], mu_c, mu2, L%op), 'ginzburg_landau_linop_gpu')
        end if
    end function

    ! ---- fused paths from Fortran: the whole step loop of a factorisation inside the engine ----------------
    ! X, U, V are engine basis handles (lk_basis_create), op an engine operator handle; H, T, B, R the caller's host arrays.
    ! A complex(dp) Fortran array IS the engine's layout (interleaved doubles): passed through c_f_pointer, no copy.
    !> Same contract as LightKrylov's `arnoldi` (src/Krylov/arnoldi.fypp:8-76): X holds (kdim+1)*blksize columns, H is
    !> ((kdim+1)*blksize x kdim*blksize); blksize > 1 runs the block factorisation (lk_arnoldi_block).
    subroutine gpu_arnoldi_rdp(op, X, H, info, kstart, kend, tol, transpose, blksize)
        type(c_ptr), intent(in) :: op, X
        real(dp), intent(inout) :: H(:, :)
        integer, intent(out) :: info
        integer, optional, intent(in) :: kstart, kend, blksize
        real(dp), optional, intent(in) :: tol
        logical, optional, intent(in) :: transpose
        integer(c_int) :: k0, k1, cinfo, p, tr
        real(c_double) :: t
        p = 1; if (present(blksize)) p = blksize
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(H, 2)/p; if (present(kend)) k1 = kend
        t = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) t = tol      ! atol_dp, Constants.f90:35
        tr = 0; if (present(transpose)) tr = merge(1, 0, transpose)
        call chk(lk_arnoldi_block(op, X, H, int(size(H, 1), c_int64_t), p, k0, k1, t, tr, cinfo), 'gpu_arnoldi_rdp')
        info = cinfo
    end subroutine

    subroutine gpu_arnoldi_cdp(op, X, H, info, kstart, kend, tol, transpose, blksize)
        type(c_ptr), intent(in) :: op, X
        complex(dp), intent(inout), target, contiguous :: H(:, :)
        integer, intent(out) :: info
        integer, optional, intent(in) :: kstart, kend, blksize
        real(dp), optional, intent(in) :: tol
        logical, optional, intent(in) :: transpose
        integer(c_int) :: k0, k1, cinfo, p, tr
        real(c_double) :: t
        real(c_double), pointer :: Hr(:)
        p = 1; if (present(blksize)) p = blksize
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(H, 2)/p; if (present(kend)) k1 = kend
        t = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) t = tol
        tr = 0; if (present(transpose)) tr = merge(1, 0, transpose)
        call c_f_pointer(c_loc(H), Hr, [2*size(H)])
        call chk(lk_arnoldi_block(op, X, Hr, int(size(H, 1), c_int64_t), p, k0, k1, t, tr, cinfo), 'gpu_arnoldi_cdp')
        info = cinfo
    end subroutine

    !> arnoldi (blksize = 1) delivered in SEGMENTS while the device runs on: on_columns(user, kfirst, klast) -- a bind(C) function of
    !> the interface lk_gpu_progress, non-zero return = stop -- is called as soon as columns kfirst..klast of H are final;
    !> seg_last = last step of each segment, ascending.  What eigs' per-step Ritz test needs (IterativeSolvers.fypp:1059-1093).
    subroutine gpu_arnoldi_segments_rdp(op, X, H, info, seg_last, on_columns, user, kstart, kend, tol, transpose)
        type(c_ptr), intent(in) :: op, X
        real(dp), intent(inout) :: H(:, :)
        integer, intent(out) :: info
        integer, intent(in) :: seg_last(:)
        procedure(lk_gpu_progress) :: on_columns
        type(c_ptr), optional, intent(in) :: user
        integer, optional, intent(in) :: kstart, kend
        real(dp), optional, intent(in) :: tol
        logical, optional, intent(in) :: transpose
        integer(c_int) :: k0, k1, cinfo, tr
        integer(c_int), allocatable :: segs(:)
        real(c_double) :: t
        type(c_ptr) :: u
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(H, 2); if (present(kend)) k1 = kend
        t = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) t = tol
        tr = 0; if (present(transpose)) tr = merge(1, 0, transpose)
        u = c_null_ptr; if (present(user)) u = user
        allocate(segs(max(size(seg_last), 1))); segs = 0; segs(:size(seg_last)) = int(seg_last, c_int)
        call chk(lk_arnoldi_segments(op, X, H, int(size(H, 1), c_int64_t), k0, k1, t, tr, segs, int(size(seg_last), c_int), &
                                     c_funloc(on_columns), u, cinfo), 'gpu_arnoldi_segments_rdp')
        info = cinfo
    end subroutine

    subroutine gpu_arnoldi_segments_cdp(op, X, H, info, seg_last, on_columns, user, kstart, kend, tol, transpose)
        type(c_ptr), intent(in) :: op, X
        complex(dp), intent(inout), target, contiguous :: H(:, :)
        integer, intent(out) :: info
        integer, intent(in) :: seg_last(:)
        procedure(lk_gpu_progress) :: on_columns
        type(c_ptr), optional, intent(in) :: user
        integer, optional, intent(in) :: kstart, kend
        real(dp), optional, intent(in) :: tol
        logical, optional, intent(in) :: transpose
        integer(c_int) :: k0, k1, cinfo, tr
        integer(c_int), allocatable :: segs(:)
        real(c_double) :: t
        real(c_double), pointer :: Hr(:)
        type(c_ptr) :: u
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(H, 2); if (present(kend)) k1 = kend
        t = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) t = tol
        tr = 0; if (present(transpose)) tr = merge(1, 0, transpose)
        u = c_null_ptr; if (present(user)) u = user
        allocate(segs(max(size(seg_last), 1))); segs = 0; segs(:size(seg_last)) = int(seg_last, c_int)
        call c_f_pointer(c_loc(H), Hr, [2*size(H)])
        call chk(lk_arnoldi_segments(op, X, Hr, int(size(H, 1), c_int64_t), k0, k1, t, tr, segs, int(size(seg_last), c_int), &
                                     c_funloc(on_columns), u, cinfo), 'gpu_arnoldi_segments_cdp')
        info = cinfo
    end subroutine

    !> lanczos_tridiagonalization (src/Krylov/lanczos.fypp:7-64) for a symmetric / Hermitian engine operator: X holds kdim+1 columns,
    !> T is (kdim+1 x kdim); every step of the call enqueued asynchronously, one host synchronisation (lk_lanczos).
    subroutine gpu_lanczos_rdp(op, X, T, info, kstart, kend, tol)
        type(c_ptr), intent(in) :: op, X
        real(dp), intent(inout) :: T(:, :)
        integer, intent(out) :: info
        integer, optional, intent(in) :: kstart, kend
        real(dp), optional, intent(in) :: tol
        integer(c_int) :: k0, k1, cinfo
        real(c_double) :: tl
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(T, 2); if (present(kend)) k1 = kend
        tl = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) tl = tol
        call chk(lk_lanczos(op, X, T, int(size(T, 1), c_int64_t), k0, k1, tl, cinfo), 'gpu_lanczos_rdp')
        info = cinfo
    end subroutine

    subroutine gpu_lanczos_cdp(op, X, T, info, kstart, kend, tol)
        type(c_ptr), intent(in) :: op, X
        complex(dp), intent(inout), target, contiguous :: T(:, :)
        integer, intent(out) :: info
        integer, optional, intent(in) :: kstart, kend
        real(dp), optional, intent(in) :: tol
        integer(c_int) :: k0, k1, cinfo
        real(c_double) :: tl
        real(c_double), pointer :: Tr(:)
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(T, 2); if (present(kend)) k1 = kend
        tl = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) tl = tol
        call c_f_pointer(c_loc(T), Tr, [2*size(T)])
        call chk(lk_lanczos(op, X, Tr, int(size(T, 1), c_int64_t), k0, k1, tl, cinfo), 'gpu_lanczos_cdp')
        info = cinfo
    end subroutine

    !> lanczos_bidiagonalization (src/Krylov/golub_kahan.fypp:7-64): U holds kdim+1 columns, V kdim (two different bases), B is
    !> (kdim+1 x kdim); both halves of every step enqueued asynchronously (lk_bidiag).
    subroutine gpu_bidiag_rdp(op, U, V, B, info, kstart, kend, tol)
        type(c_ptr), intent(in) :: op, U, V
        real(dp), intent(inout) :: B(:, :)
        integer, intent(out) :: info
        integer, optional, intent(in) :: kstart, kend
        real(dp), optional, intent(in) :: tol
        integer(c_int) :: k0, k1, cinfo
        real(c_double) :: tl
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(B, 2); if (present(kend)) k1 = kend
        tl = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) tl = tol
        call chk(lk_bidiag(op, U, V, B, int(size(B, 1), c_int64_t), k0, k1, tl, cinfo), 'gpu_bidiag_rdp')
        info = cinfo
    end subroutine

    subroutine gpu_bidiag_cdp(op, U, V, B, info, kstart, kend, tol)
        type(c_ptr), intent(in) :: op, U, V
        complex(dp), intent(inout), target, contiguous :: B(:, :)
        integer, intent(out) :: info
        integer, optional, intent(in) :: kstart, kend
        real(dp), optional, intent(in) :: tol
        integer(c_int) :: k0, k1, cinfo
        real(c_double) :: tl
        real(c_double), pointer :: Br(:)
        k0 = 1; if (present(kstart)) k0 = kstart
        k1 = size(B, 2); if (present(kend)) k1 = kend
        tl = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) tl = tol
        call c_f_pointer(c_loc(B), Br, [2*size(B)])
        call chk(lk_bidiag(op, U, V, Br, int(size(B, 1), c_int64_t), k0, k1, tl, cinfo), 'gpu_bidiag_cdp')
        info = cinfo
    end subroutine

    !> qr_no_pivoting (src/Krylov/qr.fypp:116-167) of columns j0+1 .. j0+size(R, 2) of the panel Q (j0 defaults to 0); R is p x p.
    subroutine gpu_qr_rdp(Q, R, info, tol, j0)
        type(c_ptr), intent(in) :: Q
        real(dp), intent(inout) :: R(:, :)
        integer, intent(out) :: info
        real(dp), optional, intent(in) :: tol
        integer, optional, intent(in) :: j0
        integer(c_int) :: cinfo, c0
        real(c_double) :: tl
        c0 = 0; if (present(j0)) c0 = j0
        tl = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) tl = tol
        call chk(lk_qr(Q, c0, int(size(R, 2), c_int), R, int(size(R, 1), c_int64_t), tl, cinfo), 'gpu_qr_rdp')
        info = cinfo
    end subroutine

    subroutine gpu_qr_cdp(Q, R, info, tol, j0)
        type(c_ptr), intent(in) :: Q
        complex(dp), intent(inout), target, contiguous :: R(:, :)
        integer, intent(out) :: info
        real(dp), optional, intent(in) :: tol
        integer, optional, intent(in) :: j0
        integer(c_int) :: cinfo, c0
        real(c_double) :: tl
        real(c_double), pointer :: Rr(:)
        c0 = 0; if (present(j0)) c0 = j0
        tl = 10.0_dp**(-precision(1.0_dp)); if (present(tol)) tl = tol
        call c_f_pointer(c_loc(R), Rr, [2*size(R)])
        call chk(lk_qr(Q, c0, int(size(R, 2), c_int), Rr, int(size(R, 1), c_int64_t), tl, cinfo), 'gpu_qr_cdp')
        info = cinfo
    end subroutine
end module lightkrylov_gpu
