"""Engine context: one per process = one GPU (+ optional torch.distributed process group).

PyTorch is plumbing here: it provides the current HIP stream and the RCCL communicator
(`torch.distributed` backend "nccl" is RCCL on ROCm).  All arithmetic happens in the HIP
library behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _capi


class _DevMem:
    """Zero-copy view of device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {
            "shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def row_partition(n_global: int, nranks: int, rank: int):
    """Contiguous row block of `rank`: (row0, n_local).  Every block but the last has an even
    number of rows (n_global // nranks rounded down to even); the last takes the remainder."""
    if nranks < 1 or not (0 <= rank < nranks):
        raise ValueError(f"bad rank {rank}/{nranks}")
    base = (n_global // nranks) & ~1
    row0 = base * rank
    n_local = base if rank < nranks - 1 else n_global - base * (nranks - 1)
    return row0, n_local


class Context:
    def __init__(self, device: int | None = None, stream: int | None = None, process_group=None,
                 use_torch_stream: bool = True):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        self._torch = None
        self._pg = None
        self.nranks, self.rank = 1, 0
        self.row0, self.n_global = 0, None
        self._cb = None
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        if stream is None and use_torch_stream:
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.set_device(device)
                    stream = torch.cuda.current_stream(device).cuda_stream
                    self._torch = torch
            except ImportError:
                pass
        _capi.check(self._lib.lk_init(int(device), C.c_void_p(stream or 0), C.byref(self._h)))
        self.device = int(device)
        self.stream = stream
        if process_group is not None:
            self.set_process_group(process_group)

    # -- multi-GPU: sum all-reduce of the (<= 258) reduction scalars over RCCL ------------
    def set_process_group(self, pg) -> None:
        """Route the engine's collectives (sum all-reduce of the sweep scalars, all-gather of row blocks, neighbour exchange)
        through a torch.distributed group.  backend "nccl" (= RCCL): on the device tensors themselves, on the engine's stream.
        Any other backend (gloo, mpi without device support): STAGED THROUGH HOST MEMORY by this hook -- a blocking device -> host
        copy on the engine's stream, the collective on CPU tensors, a blocking host -> device copy on the engine's stream.
        (torch's gloo backend does accept device tensors and stages them itself, on streams and events of its own; four ranks
        sharing one GPU stalled inside exactly that path -- every rank in the SAME all-reduce, same sequence number and count,
        tests/test_gpu_distributed.py, round-4 record in DESIGN.md section 6 -- so the hook does the staging where it can be seen,
        with nothing but one stream and blocking copies.)"""
        import torch
        import torch.distributed as dist
        self._torch = torch
        self._pg = pg
        self.nranks = dist.get_world_size(pg)
        self.rank = dist.get_rank(pg)
        on_device = dist.get_backend(pg) == "nccl"
        dev = f"cuda:{self.device}"
        ext_stream_cache = {}
        view_cache = {}
        host_cache = {}
        # LK_TRACE_COLLECTIVES=1: one stderr line when a data-path collective is entered and one when it returns (sequence number,
        # kind, count) -- with every rank's last lines side by side a stalled job shows whether the ranks sit in the SAME call
        trace = os.environ.get("LK_TRACE_COLLECTIVES", "0") not in ("", "0")
        seq = [0]

        def _trace(what, count, phase):
            import sys
            print(f"[lightkrylov_amd] rank {self.rank}/{self.nranks} collective #{seq[0]} {what} count={int(count)} {phase}", file=sys.stderr, flush=True)

        def _stream(stream_ptr):
            sp = int(stream_ptr or 0)
            if sp not in ext_stream_cache:
                ext_stream_cache[sp] = torch.cuda.ExternalStream(sp, device=self.device) if sp else torch.cuda.current_stream(self.device)
            return ext_stream_cache[sp]

        def _view(ptr, count):
            key = (int(ptr), int(count))
            t = view_cache.get(key)
            if t is None:
                t = torch.as_tensor(_DevMem(*key), device=dev)
                if len(view_cache) < 4096:
                    view_cache[key] = t
            return t

        def _host(tag, count):
            """host staging buffer (one per use and length, re-used).  Plain pageable memory and BLOCKING copies on the engine's
            stream: nothing here outlives the call -- no pinned blocks whose release the caching allocator would tie to the
            engine's stream by events (a first version with pinned buffers and asynchronous copies crashed in the interpreter's
            garbage collection at teardown)."""
            key = (tag, int(count))
            h = host_cache.get(key)
            if h is None:
                h = torch.empty(int(count), dtype=torch.float64)
                host_cache[key] = h
            return h

        def _src(r):
            return dist.get_global_rank(pg, r) if pg is not None and pg is not dist.group.WORLD else r

        def _allreduce(_user, dev_ptr, count, stream_ptr):
            try:
                if trace:
                    seq[0] += 1
                    _trace("all_reduce", count, "enter")
                t, st = _view(dev_ptr, count), _stream(stream_ptr)
                with torch.cuda.stream(st):
                    if on_device:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
                    else:
                        h = _host("ar", count)
                        h.copy_(t)                                   # device -> host on the engine's stream, returns when done
                        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=pg)
                        t.copy_(h)
                if trace:
                    _trace("all_reduce", count, "done")
                return 0
            except Exception as exc:  # noqa: BLE001 - must not propagate through C
                import sys
                print(f"[lightkrylov_amd] all-reduce callback failed: {exc!r}", file=sys.stderr)
                return 1

        self._cb = _capi.ALLREDUCE_FN(_allreduce)
        _capi.check(self._lib.lk_set_allreduce(self._h, self._cb, None, self.nranks, self.rank))

        def _allgather(_user, send_ptr, recv_ptr, counts, displs, nranks, stream_ptr):
            # every rank's row block of a vector to every rank (row-sharded dense / CSR matvec): one broadcast per block, which
            # every backend supports for unequal blocks
            try:
                cnts = [int(counts[r]) for r in range(int(nranks))]
                offs = [int(displs[r]) for r in range(int(nranks))]
                total = max(o + c for o, c in zip(offs, cnts)) if cnts else 0
                if trace:
                    seq[0] += 1
                    _trace("all_gather", total, "enter")
                st = _stream(stream_ptr)
                mine = cnts[self.rank]
                with torch.cuda.stream(st):
                    if on_device:
                        if mine:
                            _view(int(recv_ptr) + 8 * offs[self.rank], mine).copy_(_view(send_ptr, mine))
                        for r in range(int(nranks)):
                            if cnts[r]:
                                dist.broadcast(_view(int(recv_ptr) + 8 * offs[r], cnts[r]), src=_src(r), group=pg)
                    elif total:
                        h = _host("ag", total)
                        if mine:
                            h[offs[self.rank]:offs[self.rank] + mine].copy_(_view(send_ptr, mine))
                        for r in range(int(nranks)):
                            if cnts[r]:
                                dist.broadcast(h[offs[r]:offs[r] + cnts[r]], src=_src(r), group=pg)
                        for r in range(int(nranks)):          # (the blocks may leave gaps in recv: copy block by block)
                            if cnts[r]:
                                _view(int(recv_ptr) + 8 * offs[r], cnts[r]).copy_(h[offs[r]:offs[r] + cnts[r]])
                if trace:
                    _trace("all_gather", total, "done")
                return 0
            except Exception as exc:  # noqa: BLE001 - must not propagate through C
                import sys
                print(f"[lightkrylov_amd] all-gather callback failed: {exc!r}", file=sys.stderr)
                return 1

        self._ag_cb = _capi.ALLGATHER_FN(_allgather)
        _capi.check(self._lib.lk_set_allgather(self._h, self._ag_cb, None))

        def _halo(_user, send_lo, send_hi, recv_lo, recv_hi, count, stream_ptr):
            # nearest-neighbour exchange of the stencil operators through collectives every backend has: all ranks gather
            # every rank's two edge blocks and keep their neighbours' (the native communicator uses ncclSend / ncclRecv)
            try:
                cnt = int(count)
                if trace:
                    seq[0] += 1
                    _trace("halo", cnt, "enter")
                st = _stream(stream_ptr)
                with torch.cuda.stream(st):
                    mine = torch.zeros(2 * cnt, dtype=torch.float64, device=dev) if on_device else _host("halo", 2 * cnt)
                    if not on_device:
                        mine.zero_()
                    if send_lo:
                        mine[:cnt].copy_(_view(send_lo, cnt))
                    if send_hi:
                        mine[cnt:].copy_(_view(send_hi, cnt))
                    every = [torch.empty_like(mine) for _ in range(self.nranks)]
                    dist.all_gather(every, mine, group=pg)
                    if recv_lo and self.rank > 0:
                        _view(recv_lo, cnt).copy_(every[self.rank - 1][cnt:])
                    if recv_hi and self.rank + 1 < self.nranks:
                        _view(recv_hi, cnt).copy_(every[self.rank + 1][:cnt])
                if trace:
                    _trace("halo", cnt, "done")
                return 0
            except Exception as exc:  # noqa: BLE001 - must not propagate through C
                import sys
                print(f"[lightkrylov_amd] halo-exchange callback failed: {exc!r}", file=sys.stderr)
                return 1

        self._halo_cb = _capi.HALO_FN(_halo)
        _capi.check(self._lib.lk_set_halo_exchange(self._h, self._halo_cb, None))

    # -- multi-GPU, native: ncclAllReduce issued by the library itself (no interpreter in the path) ------
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: a fresh RCCL unique id (LK_COMM_ID_BYTES opaque bytes) to ship to every rank."""
        buf = C.create_string_buffer(_capi.LK_COMM_ID_BYTES)
        _capi.check(_capi.load().lk_comm_get_unique_id(buf))
        return buf.raw

    def init_native_comm(self, nranks: int, rank: int, unique_id: bytes) -> None:
        """Collective over the `nranks` processes sharing a row-sharded basis: ncclCommInitRank on this
        context's device; afterwards every sweep's reduction scalars are summed by ncclAllReduce on the
        context's stream (lk_comm_init_rank)."""
        if len(unique_id) != _capi.LK_COMM_ID_BYTES:
            raise ValueError("unique_id must be LK_COMM_ID_BYTES bytes")
        buf = C.create_string_buffer(unique_id, _capi.LK_COMM_ID_BYTES)
        _capi.check(self._lib.lk_comm_init_rank(self._h, int(nranks), int(rank), buf))
        self.nranks, self.rank = int(nranks), int(rank)
        self._cb = None

    @staticmethod
    def native_comm_available() -> bool:
        """LOCAL check (no collective, no GPU work): librccl and every entry point the communicator needs resolve in this
        process (lk_comm_available)."""
        return _capi.load().lk_comm_available() == 0

    def init_native_comm_from_process_group(self, pg=None) -> bool:
        """Bootstrap the native communicator through an existing torch.distributed group -- COLLECTIVELY DECIDED, so that the
        ranks can never disagree on the reduction route (a rank on the callback route and a rank on the native one would each
        wait for the other in the first sweep):
          1. every rank checks locally that librccl resolves and the group takes the MINIMUM of the flags: when it is missing
             anywhere, nothing is installed anywhere and every rank returns False (the caller picks another route, the same on
             all ranks);
          2. rank 0's unique id is broadcast over `pg` (torch only carries the 128 bootstrap bytes; the data path is the
             library's) and every rank enters ncclCommInitRank;
          3. the group takes the minimum of the outcomes: when ncclCommInitRank failed on any rank, every rank destroys what it
             built and raises -- the same error everywhere.
        A rank that cannot even enter step 2 must leave the job (a non-zero exit: the launcher tears the others down); it cannot
        be waited for, the others are inside the communicator's bootstrap."""
        import torch
        import torch.distributed as dist
        nranks, rank = dist.get_world_size(pg), dist.get_rank(pg)
        on_device = dist.get_backend(pg) == "nccl"
        dev = torch.device(f"cuda:{self.device}") if on_device else torch.device("cpu")

        def agree(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
            return bool(int(flag.item()))

        have = self.native_comm_available()
        if not agree(have):
            if not have:
                import sys
                print(f"[lightkrylov_amd] rank {rank}: librccl unavailable ({_capi.load().lk_last_error().decode()})", file=sys.stderr)
            return False
        payload = [self.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(payload, src=dist.get_global_rank(pg, 0) if pg is not None else 0, group=pg,
                                   device=dev if on_device else None)
        err = None
        try:
            self.init_native_comm(nranks, rank, payload[0])
        except Exception as exc:  # noqa: BLE001 - the outcome is agreed on below, then raised on every rank
            err = exc
        if not agree(err is None):
            if err is None:
                self.destroy_native_comm()
            raise RuntimeError(f"native RCCL communicator failed on at least one rank (rank {rank}: {err!r})")
        self._pg = pg
        return True

    def destroy_native_comm(self) -> None:
        _capi.check(self._lib.lk_comm_destroy(self._h))
        self.nranks, self.rank = 1, 0

    def set_halo_exchange(self, fn) -> None:
        """Install a host-provided nearest-neighbour exchange (`_capi.HALO_FN`) for the stencil operators; the native
        communicator installs its own (ncclSend / ncclRecv)."""
        self._halo_cb = fn
        _capi.check(self._lib.lk_set_halo_exchange(self._h, fn, None))

    def set_allgather(self, fn) -> None:
        """Install a host-provided all-gather of row blocks (`_capi.ALLGATHER_FN`) for the row-sharded dense / CSR operators; the
        native communicator and `set_process_group` install their own."""
        self._ag_cb = fn
        _capi.check(self._lib.lk_set_allgather(self._h, fn, None))

    def set_partition(self, row0: int, n_global: int) -> None:
        _capi.check(self._lib.lk_set_partition(self._h, int(row0), int(n_global)))
        self.row0, self.n_global = int(row0), int(n_global)

    def set_tuning(self, key: str, value: int) -> None:
        _capi.check(self._lib.lk_set_tuning(self._h, key.encode(), int(value)))

    def lazy_stats(self):
        """(dot memo hits, batched dot sweeps, queued axpbys, queue flushes) of the lazy per-object path."""
        out = (C.c_int64 * 4)()
        _capi.check(self._lib.lk_lazy_stats(self._h, out))
        return tuple(out)

    def engine_stream(self) -> int:
        """The HIP stream every engine call of this context runs on (lk_context_info)."""
        dev, st = C.c_int(), C.c_void_p()
        _capi.check(self._lib.lk_context_info(self._h, C.byref(dev), C.byref(st)))
        return int(st.value or 0)

    def torch_stream(self):
        """`with ctx.torch_stream(): ...` makes torch ops inside run ON the engine's stream, i.e. ordered with every engine
        call before and after -- what code touching `dense_vector_gpu.as_torch()` views needs.  (A context created from
        torch's DEFAULT stream gets a stream of its own, so torch ops outside this block are NOT ordered with the engine.)"""
        import torch
        if getattr(self, "_ext_stream", None) is None:
            self._ext_stream = torch.cuda.ExternalStream(self.engine_stream(), device=self.device)
        return torch.cuda.stream(self._ext_stream)

    def lazy_fusion_stats(self):
        """(fused update+dot sweeps, pending updates applied as plain panel updates, virtual temporaries dropped
        unwritten, virtual temporaries written after all) -- see lk_lazy_fusion_stats in the header."""
        out = (C.c_int64 * 4)()
        _capi.check(self._lib.lk_lazy_fusion_stats(self._h, out))
        return tuple(out)

    def resident_stats(self):
        """(single-launch Gram-Schmidt steps enqueued, launches that gave up, launches that kept the panel in registers) --
        see lk_resident_stats in the header."""
        out = (C.c_int64 * 3)()
        _capi.check(self._lib.lk_resident_stats(self._h, out))
        return tuple(out)

    def resident_phase_us(self):
        """Durations in microseconds of the last single launch as block 0 saw it: (phase 1, sum 1, phase 2, sum 2, phase 3, sum 3, scale)."""
        out = (C.c_int64 * 8)()
        _capi.check(self._lib.lk_resident_phase_ticks(self._h, out))
        return tuple((out[i + 1] - out[i]) / 100.0 for i in range(7))

    def lazy_speculation_stats(self):
        """(anticipated first-pass sweeps, of which unused) -- see lk_lazy_speculation_stats in the header."""
        out = (C.c_int64 * 2)()
        _capi.check(self._lib.lk_lazy_speculation_stats(self._h, out))
        return tuple(out)

    def sync(self) -> None:
        _capi.check(self._lib.lk_sync(self._h))

    def profile_enable(self, on: bool = True) -> None:
        _capi.check(self._lib.lk_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self) -> None:
        _capi.check(self._lib.lk_profile_reset(self._h))

    def profile_get(self, tag: str):
        cnt, ms, by = C.c_int64(), C.c_double(), C.c_double()
        _capi.check(self._lib.lk_profile_get(self._h, tag.encode(), C.byref(cnt), C.byref(ms), C.byref(by)))
        return cnt.value, ms.value, by.value

    def close(self) -> None:
        if self._h:
            self._lib.lk_finalize(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_default: Context | None = None


def default_context() -> Context:
    global _default
    if _default is None:
        _default = Context()
    return _default
