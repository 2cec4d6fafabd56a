"""Iterative solvers that call the hot path -- mirror of src/IterativeSolvers (gmres, eigs).

Control flow, defaults, `info` and metadata follow the reference line by line; the small
host problems (Givens / trtrs / geev / gees / trsen on <= 129 x 128 matrices) are LAPACK
calls on the host exactly where the reference makes them (scipy's LAPACK here).  All O(n)
work goes through the vector type, i.e. the HIP engine for `dense_vector_gpu`.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.linalg import lapack as _lapack

import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

from . import _hostlapack
from .constants import atol_dp, rtol_dp
from .krylov import krylov_schur_host_part  # noqa: E402,F401  (re-exported for the pipelined eigs cycle)
from .krylov import _engine_linop, arnoldi, bidiagonalization, double_gram_schmidt_step, krylov_schur, lanczos
from .linops import abstract_linop
from .outputs import eigs_output, write_results
from .vectors import abstract_vector, copy, dense_vector_gpu, krylov_basis_gpu, linear_combination, zero_basis


# ------------------------------------------------------------------------------------------
@dataclass
class gmres_dp_opts:
    """IterativeSolvers.fypp:141-151"""
    kdim: int = 30
    maxiter: int = 10
    if_print_metadata: bool = False
    sanity_check: bool = True


@dataclass
class gmres_dp_metadata:
    """IterativeSolvers.fypp:153-170"""
    n_iter: int = 0
    n_inner: int = 0
    n_outer: int = 0
    res: list = field(default_factory=list)
    converged: bool = False
    info: int = 0


@dataclass
class cg_dp_opts:
    """IterativeSolvers.fypp (cg_dp_opts)"""
    maxiter: int = 100
    if_print_metadata: bool = False


@dataclass
class cg_dp_metadata:
    """IterativeSolvers.fypp (cg_dp_metadata)"""
    n_iter: int = 0
    res: list = field(default_factory=list)
    converged: bool = False
    info: int = 0


def eig(Hk: np.ndarray):
    """eig(A, vecs, vals): LAPACK geev, right eigenvectors in LAPACK layout (real pairs are NOT
    combined).  src/Utilities/submodule_utility_functions.fypp:55-85"""
    if Hk.dtype == np.float64:
        wr, wi, _vl, vr, info = _lapack.dgeev(np.asfortranarray(Hk), compute_vl=0, compute_vr=1)
        vals = wr + 1j * wi
    else:
        vals, _vl, vr, info = _lapack.zgeev(np.asfortranarray(Hk), compute_vl=0, compute_vr=1)
    if info != 0:
        raise RuntimeError(f"GEEV failed, info={info}")
    return vr, vals


def apply_givens_rotation(h: np.ndarray, c: np.ndarray, s: np.ndarray) -> None:
    """submodule_utility_functions.fypp:173-204.  h has k+1 entries; c, s have k."""
    k = h.size - 1
    if h.dtype == np.float64:
        for j in range(k - 1):                    # lasr("L","V","F"): apply the previous rotations
            t = h[j + 1]
            h[j + 1] = c[j] * t - s[j] * h[j]
            h[j] = s[j] * t + c[j] * h[j]
        cc, ss, r = _lapack.dlartg(h[k - 1], h[k])
        c[k - 1], s[k - 1] = cc, ss
        h[k - 1], h[k] = r, 0.0
    else:
        for i in range(k - 1):
            t = c[i] * h[i] + s[i] * h[i + 1]
            h[i + 1] = -s[i] * h[i] + c[i] * h[i + 1]
            h[i] = t
        g = np.array([h[k - 1], h[k]])
        g = g / np.sqrt(np.sum(np.abs(g) ** 2))   # givens_rotation: x / norm(x, 2)
        c[k - 1], s[k - 1] = g[0], g[1]
        h[k - 1] = c[k - 1] * h[k - 1] + s[k - 1] * h[k]
        h[k] = 0.0


def _new_basis(proto: abstract_vector, ncols: int):
    """allocate(V(ncols), source=proto); zero_basis(V)"""
    if isinstance(proto, dense_vector_gpu):
        return krylov_basis_gpu(proto.basis.n_local, ncols, proto.dtype, proto.basis.ctx)
    V = [proto.zeros_like() for _ in range(ncols)]
    zero_basis(V)
    return V


def _dtype_of(v: abstract_vector):
    return getattr(v, "dtype", np.dtype(np.float64))


# ------------------------------------------------------------------------------------------
def gmres(A: abstract_linop, b: abstract_vector, x: abstract_vector, rtol: float = rtol_dp,
          atol: float = atol_dp, preconditioner=None, options: gmres_dp_opts | None = None,
          transpose: bool = False, meta: gmres_dp_metadata | None = None) -> int:
    """Restarted GMRES(kdim).  src/IterativeSolvers/GMRES/gmres.fypp:65-255.
    x is the initial guess and is overwritten by the solution.  Returns info = +n_iter if
    converged, -n_iter otherwise (:234-238); `meta` (if given) receives the residual history."""
    opts = options or gmres_dp_opts()
    kdim, maxiter = opts.kdim, opts.maxiter
    dt = _dtype_of(b)
    tol = atol + rtol * b.norm()                                                   # :106
    wrk = b.zeros_like()
    V = _new_basis(b, kdim + 1)
    H = np.zeros((kdim + 1, kdim), dtype=dt, order="F")
    m = gmres_dp_metadata()
    A.reset_counter(transpose, "gmres%init")
    mv = A.apply_rmatvec if transpose else A.apply_matvec

    while (not m.converged) and m.n_outer <= maxiter:                              # :131
        H[...] = 0
        zero_basis(V)
        if x.norm() != 0.0:
            mv(x, V[0])                                                            # :134-140
        V[0].sub(b)
        V[0].chsgn()                                                               # :141
        e = np.zeros(kdim + 1, dtype=dt)
        beta = V[0].norm()
        e[0] = beta
        V[0].scal(1.0 / beta)
        c = np.zeros(kdim, dtype=dt)
        s = np.zeros(kdim, dtype=dt)
        if m.n_outer == 0:
            m.res = [abs(beta)]
        k = 0
        fused = (preconditioner is None and isinstance(V, krylov_basis_gpu) and isinstance(A, _engine_linop)
                 and H.flags.f_contiguous and kdim >= 2 and _GMRES_FUSED)
        if fused:
            # The inner cycle as ONE call into the engine (lk_arnoldi_segments, round 5): an inner iteration of gmres IS an Arnoldi step --
            # matvec, double Gram-Schmidt with beta = H(:k, k), H(k+1, k) = the norm, scale unless it is below tol (:155-172) -- so the
            # kdim steps are enqueued back to back and the engine reports every column of H the moment it is final; the Givens rotations
            # and the residual test (:178-196) run here while the device is already on the next steps, and a converged residual stops
            # the factorisation (at most 24 steps beyond it have run: they touched columns of V the solution does not use).  One host
            # round trip per CYCLE instead of one per step (which left the device idle for ~0.1 ms of every ~1 ms step at configs[2]).
            last = [0]

            def on_columns(kfirst, klast):
                for kk in range(kfirst, klast + 1):
                    apply_givens_rotation(H[:kk + 1, kk - 1], c[:kk], s[:kk])      # :178
                    e[kk] = -s[kk - 1] * e[kk - 1]
                    e[kk - 1] = c[kk - 1] * e[kk - 1]                              # :180
                    m.n_iter += 1
                    m.n_inner += 1
                    m.res.append(abs(e[kk]))
                    last[0] = kk
                    if abs(e[kk]) < tol:
                        m.converged = True
                        return True
                return False
            arnoldi(A, V, H, 1, kdim, tol, transpose, _segments=list(range(1, kdim + 1)), _progress=on_columns)
            k = last[0]
            beta = abs(e[k])
        # (fused and not converged with steps left: H(k+1, k) fell below tol with the residual still above it -- the reference goes on with the
        # unscaled vector, :171-172; so do the remaining steps, one by one)
        for k in range(1 if not fused else (kdim + 1 if m.converged else last[0] + 1), kdim + 1):
            if preconditioner is None and isinstance(V, krylov_basis_gpu):
                mv(V[k - 1], V[k])             # wrk is only a copy of V(k) for the preconditioner to overwrite (:155-165)
            else:
                copy(wrk, V[k - 1])                                                # :155  wrk = V(k)
                if preconditioner is not None:
                    preconditioner.apply(wrk, k, beta, tol)
                mv(wrk, V[k])                                                      # :161-165
            hcol = np.zeros(k, dtype=dt)
            norms: list = []
            if isinstance(V, krylov_basis_gpu):
                double_gram_schmidt_step(V[k], V[:k], False, hcol, _norms=norms)   # :167-168
                hk1 = norms[2]                                                     # :171 (norm comes out of sweep 3)
            else:
                double_gram_schmidt_step(V[k], V[:k], False, hcol)
                hk1 = V[k].norm()
            H[:k, k - 1] = hcol
            H[k, k - 1] = hk1
            if abs(H[k, k - 1]) > tol:
                V[k].scal(1.0 / hk1)                                               # :172
            apply_givens_rotation(H[:k + 1, k - 1], c[:k], s[:k])                  # :178
            e[k] = -s[k - 1] * e[k - 1]
            e[k - 1] = c[k - 1] * e[k - 1]                                         # :180
            beta = abs(e[k])
            m.n_iter += 1
            m.n_inner += 1
            m.res.append(abs(beta))
            if abs(beta) < tol:
                m.converged = True
                break
        k = min(k, kdim)
        trtrs = _lapack.dtrtrs if dt == np.float64 else _lapack.ztrtrs
        yk, tinfo = trtrs(np.asfortranarray(H[:k, :k]), e[:k].copy(), lower=0, trans=0, unitdiag=0)  # :199-200
        if tinfo != 0:
            raise RuntimeError(f"TRTRS failed, info={tinfo}")
        dx = linear_combination(V[:k], yk)                                         # :201
        if preconditioner is not None:
            preconditioner.apply(dx)
        x.add(dx)
        mv(x, V[0])                                                                # :205-210
        V[0].sub(b)
        V[0].chsgn()
        beta = V[0].norm()
        if abs(beta) > 0.0:
            V[0].scal(1.0 / beta)
        m.n_iter += 1
        m.n_outer += 1
        m.res.append(abs(beta))
        if abs(beta) < tol:
            m.converged = True
            break
    info = m.n_iter if m.converged else -m.n_iter                                  # :234-238
    m.info = info
    if meta is not None:
        meta.__dict__.update(m.__dict__)
    A.reset_counter(transpose, "gmres%post")
    return info


# ------------------------------------------------------------------------------------------
_GMRES_FUSED = True          # gmres' inner cycle as one lk_arnoldi_segments call (False: one engine round trip per inner step, the round-1..4 schedule)
_EIGS_SEGMENT = 16          # Arnoldi steps per asynchronous device batch of the pipelined eigs cycle
_eigs_trace = None          # diagnostic (tools/profile_eigs_cycle.py): a list that receives (label, perf_counter()) marks of a cycle


def _mark(label: str) -> None:
    if _eigs_trace is not None:
        import time
        _eigs_trace.append((label, time.perf_counter()))


def _tapered_segments(kstart: int, kdim: int, seg: int = 0):
    """Step ranges [(a, b), ...] covering kstart..kdim: the first half as one segment, then segments of `seg` steps, the LAST `seg` steps
    split 8, 4, 2, 1, 1 (halves down to single steps), so that the host work that can only start after a segment's last column shrinks
    towards the end of the cycle."""
    seg = seg or _EIGS_SEGMENT
    out, a = [], kstart
    tail_from = max(kstart, kdim - seg + 1)
    # the first half of the range in ONE segment: its eigenproblems are the small ones (a geev of order <= kdim / 2 costs an eighth of the
    # last one) and are all done long before the cycle's device work ends, and every segment boundary costs the device ~0.4 ms of idling
    head = kstart + (kdim - kstart + 1) // 2 - 1
    if head - a + 1 > seg and head < tail_from:
        out.append((a, head))
        a = head + 1
    while a < tail_from:
        b = min(a + seg - 1, tail_from - 1)
        out.append((a, b))
        a = b + 1
    left = kdim - a + 1
    while left > 0:
        take = max(1, left // 2)
        out.append((a, a + take - 1))
        a += take
        left -= take
    return out


def _schur_then_final_eig(Hc: np.ndarray, kdim: int, select_eigs, final_eig: bool):
    """Host-thread job of the pipelined eigs cycle: the small-matrix half of krylov_schur on a copy of the complete H and -- when this is
    the cycle after which eigs returns -- the `eig` of the RESTARTED H(1:kdim, 1:kdim) that eigs ends with (IterativeSolvers.fypp:1115),
    formed exactly as krylov_schur leaves it (BaseKrylov.fypp:807-829).  Returns (host_part, None | (that matrix, eig's result)); the
    caller uses the second only if the H it then holds equals that matrix bit for bit."""
    host_part = krylov_schur_host_part(Hc, kdim, select_eigs)
    _mark("schur host part done (worker)")
    if not final_eig:
        return host_part, None
    T, Tk, Z, n = host_part
    m = Hc.shape[1]
    Hc[:m, :] = T
    Hc[:kdim, :] = Tk
    b = Hc[kdim, :] @ Z                                                             # :827
    Hc[n, :] = b
    Hc[n + 1:, :] = 0
    Hc[:, n:] = 0
    Hk = np.array(Hc[:kdim, :kdim], order="F", copy=True)
    out = eig(Hk)
    _mark("final eig ahead done (worker)")
    return host_part, (Hk, out)
_pools: dict = {}


def _pool(name: str, nthreads: int) -> ThreadPoolExecutor:
    """Host worker threads, created once (starting 32 threads costs ~15 ms: as much as the Schur step of a cycle)."""
    key = (name, nthreads)
    if key not in _pools:
        _pools[key] = ThreadPoolExecutor(nthreads, thread_name_prefix="lk_" + name)
    return _pools[key]


@_hostlapack.small_problems
def eigs(A: abstract_linop, X, x0: abstract_vector | None = None, kdim: int | None = None,
         tolerance: float = rtol_dp, transpose: bool = False, write_intermediate: bool = False,
         max_restarts: int | None = None, pipelined: bool | None = None):
    """Krylov-Schur eigensolver for the leading len(X) eigenpairs.
    src/IterativeSolvers/IterativeSolvers.fypp:972-1143.
    X (sequence / basis of nev vectors) receives the eigenvectors.
    Returns (eigvals[nev] complex, residuals[nev], info = number of Arnoldi steps).
    (`write_intermediate` defaults to False here -- the reference's default is .true. -- because the per-step
    text dump is file I/O outside the path; when True it writes `eigs_output.txt` exactly like :1091.
    `max_restarts` is an engine extra: the reference loops until `nev` pairs converge, however long.
    `pipelined` (engine extra; None = whenever possible): see the comment at the cycle loop -- same results.)"""
    nev = len(X)
    kdim_ = 4 * nev if kdim is None else kdim                                      # :1023
    proto = X[0]
    dt = _dtype_of(proto)
    Xwrk = _new_basis(proto, kdim_ + 1)                                            # :1032-1034
    if x0 is not None:
        copy(Xwrk[0], x0)
        Xwrk[0].scal(1.0 / x0.norm())                                              # :1036-1038
    else:
        Xwrk[0].rand(True)                                                         # :1040
    H = np.zeros((kdim_ + 1, kdim_), dtype=dt, order="F")
    res = np.zeros(kdim_)
    kstart, conv, niter, k = 1, 0, 0, 0

    def median_selector(lam):                                                      # :1137-1142
        return np.abs(lam) > np.median(np.abs(lam))

    def ritz_test(k, Hk=None):
        """Ritz values of H(1:k, 1:k) and their residuals |H(k+1,k) * last eigenvector component|   :1065-1082
        (Hk: a private copy of H(1:k+1, 1:k) -- the pipelined cycle hands every test its own, so that a test still running when the
        cycle is cut short never reads H while the restart rewrites it)"""
        Hk = H if Hk is None else Hk
        vecs, vals = _hostlapack.geev(Hk[:k, :k])                                  # :1065 (same LAPACK routine as `eig`)
        beta = Hk[k, k - 1]
        r = np.empty(k)
        if dt == np.complex128:
            r[:] = np.abs(beta * vecs[k - 1, :k])                                  # :1071
        else:
            for i in range(k):                                                     # :1073-1082
                if vals[i].imag > 0:
                    alpha = abs(complex(vecs[k - 1, i], vecs[k - 1, i + 1]))
                elif vals[i].imag < 0:
                    alpha = abs(complex(vecs[k - 1, i - 1], vecs[k - 1, i]))
                else:
                    alpha = abs(vecs[k - 1, i])
                r[i] = abs(beta * alpha)
        return vals, r

    # Whole-cycle pipeline (engine extra, result-identical).  The reference alternates ONE Arnoldi step with ONE geev of
    # H(1:k, 1:k) and stops at the first k with `nev` converged Ritz pairs (:1059-1093).  The Hessenberg columns do not
    # depend on when they are looked at, so here the cycle's steps are enqueued in one asynchronous lk_arnoldi call, and the
    # per-step tests -- independent once H is known -- run afterwards on several host threads, in step order, stopping at
    # the first k the reference would have stopped at.  Steps computed beyond it only touched eigs' private work basis.
    can_pipeline = isinstance(Xwrk, krylov_basis_gpu) and isinstance(A, _engine_linop) and not write_intermediate
    multi_rank = getattr(getattr(Xwrk, "ctx", None), "nranks", 1) > 1
    pipelined = can_pipeline and (_hostlapack.threaded() if pipelined is None else bool(pipelined))
    nthreads = max(1, min(32, os.cpu_count() or 1))

    restarts = 0
    while conv < nev:
        if max_restarts is not None and restarts > max_restarts:
            break
        restarts += 1
        k_from, stopped = kstart, False
        schur_ahead = None
        eig_ahead = None
        if pipelined and kstart <= kdim_:
            # The whole cycle is ONE call into the engine (lk_arnoldi_segments, on a helper thread; ctypes drops the interpreter lock for its
            # duration): the steps are enqueued back to back and the engine reports the columns of H segment by segment while the device runs
            # on -- no idle gap on the device between segments (round 5; one blocking lk_arnoldi call per segment cost ~0.4 ms each).  The moment
            # a segment's columns exist the Ritz test of every step in it is handed to the host pool, and this thread collects the tests IN
            # STEP ORDER and stops where the reference would have stopped; the engine is then told to enqueue nothing more (it keeps the
            # device at most 24 steps ahead of the segment it reports).  The tests of the last columns can only start when the device has
            # finished, so the segments TAPER towards the end of the cycle (..., 16, 8, 4, 2, 1, 1 steps): when the final column arrives one
            # or two `geev`s are still to be started instead of sixteen.
            bounds = _tapered_segments(kstart, kdim_)
            pool, device, feeder = _pool("geev", nthreads), _pool("device", 1), _pool("feeder", 1)
            tests: dict = {}
            last_cycle = max_restarts is not None and restarts > max_restarts        # the while loop ends after this cycle's restart
            ahead: dict = {}
            arrived: queue.Queue = queue.Queue()                                   # (kfirst, klast) ranges whose tests have been submitted; None = cycle over
            enough = threading.Event()                                             # set by this thread at the step the reference stops at

            def on_progress(kfirst, klast):
                # columns kfirst..klast of H are final (called by the engine on the device helper thread while the device runs on): the tests
                # are handed to the pool by the FEEDER thread -- sixteen `submit`s would hold the engine's loop for half a millisecond
                feeder.submit(feed, kfirst, klast)
                # true: enqueue nothing more.  `enough` is raised by the collecting thread WHEN its Ritz tests finish -- a matter of host
                # timing -- so on a row-sharded context the ranks would stop enqueueing at different steps and their all-reduces would no
                # longer pair up (round-5 advisor): there the cycle always runs to kdim; the extra steps touch eigs' private basis only.
                return enough.is_set() and not multi_rank

            def feed(kfirst, klast):
                _mark(f"columns {kfirst}..{klast} delivered")
                if klast == kdim_:
                    # H is complete: unless one of the Ritz tests still to come stops the cycle early, the restart below factors
                    # exactly this H -- start its small-matrix half (gees, selector, trsen) now, on a spare host thread, beside
                    # the last tests (same LAPACK calls on the same data; discarded on an early stop) ...
                    ahead["schur"] = pool.submit(_schur_then_final_eig, H.copy(order="F"), kdim_, median_selector, last_cycle)
                for kk in range(kfirst, klast + 1):
                    tests[kk] = pool.submit(ritz_test, kk, H[:kk + 1, :kk].copy(order="F"))
                arrived.put((kfirst, klast))

            def run_cycle():
                try:
                    return arnoldi(A, Xwrk, H, kstart, kdim_, atol_dp, transpose, _segments=[b_ for _a, b_ in bounds], _progress=on_progress)
                finally:
                    _mark("device cycle done")
                    feeder.submit(arrived.put, None)                               # behind every delivery of this cycle

            with _hostlapack.blas_threads(1):
                cycle = device.submit(run_cycle)
                k = kstart - 1
                collected = False
                try:
                    while True:
                        item = arrived.get()
                        if item is None:
                            break
                        for k in range(item[0], item[1] + 1):
                            _vals, r = tests[k].result()
                            res[:k] = r
                            niter += 1
                            conv = int(np.count_nonzero(res[:k] < tolerance))      # :1087
                            if conv >= nev:
                                stopped = True
                                break
                        if stopped:
                            enough.set()
                            break
                    collected = True
                finally:
                    if not collected:
                        # a Ritz test raised (geev failure, KeyboardInterrupt): the device thread is still inside lk_arnoldi_segments, writing H
                        # and Xwrk on the context's stream -- tell it to stop, wait for it and for the feeder, drop the tests; only then let
                        # the exception travel (round-5 advisor: the shared one-thread `device` pool stayed occupied otherwise)
                        enough.set()
                        try:
                            cycle.result()
                        except BaseException:  # noqa: BLE001 - the original exception is the one to report
                            pass
                        feeder.submit(lambda: None).result()
                        for tf in tests.values():
                            tf.cancel()
                ainfo = cycle.result()                                             # (steps the device ran beyond an early stop only touched
                kdone = ainfo if ainfo > 0 else kdim_                              #  eigs' private work basis; they are wiped below)
                feeder.submit(lambda: None).result()                               # every delivery has been handed to the pool
                for kk, tf in list(tests.items()):                                 # tests beyond the stop: never started, or left to finish unread
                    if kk > k:
                        tf.cancel()
                _mark(f"tests collected up to step {k}")
                if "schur" in ahead:                                               # (collected while the BLAS libraries' own threading is still off)
                    schur_ahead, eig_ahead = ahead["schur"].result()
                    _mark("schur (+ final eig) ahead collected")
            if stopped and k < kdone:
                # put the work arrays into the state the reference is in when it leaves the loop at step k (:1093):
                # krylov_schur below acts on ALL of H and Xwrk
                H[:, k:] = 0
                zero_basis(Xwrk[k + 1:])
            k_from = kdone + 1                                                     # (breakdown without convergence: go on step by step)
        if not stopped:
            for k in range(k_from, kdim_ + 1):
                arnoldi(A, Xwrk, H, kstart=k, kend=k, transpose=transpose)        # :1059
                vals, r = ritz_test(k)
                res[:k] = r
                niter += 1
                conv = int(np.count_nonzero(res[:k] < tolerance))                  # :1087
                if write_intermediate:
                    write_results(eigs_output, vals[:k], res[:k], tolerance)       # :1091 (sorts res(:k) in place)
                if conv >= nev:
                    break
        host_part = schur_ahead if (schur_ahead is not None and not stopped) else None   # (an early stop changed H: factor that one)
        kstart = krylov_schur(Xwrk, H, median_selector, _host_part=host_part) + 1  # :1100
        _mark("krylov_schur done")
        if host_part is None:
            eig_ahead = None
    k = min(k, kdim_)
    if eig_ahead is not None and k == kdim_ and np.array_equal(eig_ahead[0], H[:k, :k]):
        vecs, vals = eig_ahead[1]                                                  # :1115, computed ahead on exactly this matrix
    else:
        vecs, vals = eig(H[:k, :k])                                                # :1115
    vals_f = np.zeros(kdim_, dtype=np.complex128)
    vals_f[:k] = vals
    vecs_f = np.zeros((kdim_, kdim_), dtype=vecs.dtype)
    vecs_f[:k, :k] = vecs
    idx = np.argsort(-np.abs(vals_f), kind="stable")                               # :1118-1120
    vals_f, vecs_f, res_f = vals_f[idx], vecs_f[:, idx], res[idx]
    # eigenvectors X(i) = sum_j eigvecs(j, i) Xwrk(j)                                :1127-1132
    coef = np.asfortranarray(vecs_f[:k, :nev].astype(dt))
    _mark("final eig done")
    Y = linear_combination(Xwrk[:k], coef)
    copy(X, Y)
    _mark("eigenvectors done")
    return vals_f[:nev].copy(), res_f[:nev].copy(), niter


# ------------------------------------------------------------------------------------------
# The other solver families of the reference: callers of the same primitives, restated for completeness of the
# host-side mirror (through the Fortran plugin the reference's own drivers run unchanged).
def cg(A: abstract_linop, b: abstract_vector, x: abstract_vector, rtol: float = rtol_dp, atol: float = atol_dp,
       preconditioner=None, options: cg_dp_opts | None = None, meta: cg_dp_metadata | None = None) -> int:
    """Conjugate gradient for symmetric / Hermitian positive definite operators.
    src/IterativeSolvers/CG/CG.fypp:98-200.  x is the initial guess and is overwritten by the solution.
    Returns info = +n_iter if converged, -n_iter otherwise."""
    opts = options or cg_dp_opts()
    tol = atol + rtol * b.norm()                                                   # :110
    r, p, Ap = b.zeros_like(), b.zeros_like(), b.zeros_like()                      # allocate(..., mold=b); zero()   :113-121
    m = cg_dp_metadata()
    A.reset_counter(False, "cg%init")
    if x.norm() > 0:
        A.apply_matvec(x, r)                                                       # :130
    r.sub(b)
    r.chsgn()                                                                      # r = b - A x   :131
    z = None
    if preconditioner is not None:
        z = r.zeros_like(); copy(z, r); preconditioner.apply(z); copy(p, z)        # z = r ; apply ; p = z   :134
        rr_old = r.dot(z)
    else:
        copy(p, r)                                                                 # p = r   :137
        rr_old = r.dot(r)
    m.res = [float(np.sqrt(abs(rr_old)))]
    for _ in range(opts.maxiter):
        A.apply_matvec(p, Ap)                                                      # :146
        alpha = rr_old / p.dot(Ap)                                                 # :148
        x.axpby(alpha, p, 1.0)                                                     # :150
        r.axpby(-alpha, Ap, 1.0)                                                   # :152
        if preconditioner is not None:
            copy(z, r); preconditioner.apply(z)
            rr_new = r.dot(z)
        else:
            rr_new = r.dot(r)
        residual = float(np.sqrt(abs(rr_new)))
        m.n_iter += 1
        m.res.append(residual)
        if residual < tol:
            m.converged = True
            break
        beta = rr_new / rr_old                                                     # :174
        p.axpby(1.0, z if preconditioner is not None else r, beta)                 # :178-180
        rr_old = rr_new
    info = m.n_iter if m.converged else -m.n_iter
    m.info = info
    if meta is not None:
        meta.__dict__.update(m.__dict__)
    A.reset_counter(False, "cg%post")
    return info


@_hostlapack.small_problems
def eighs(A: abstract_linop, X, x0: abstract_vector | None = None, kdim: int | None = None,
          tolerance: float = rtol_dp, write_intermediate: bool = False, pipelined: bool | None = None):
    """Lanczos eigensolver for the leading len(X) eigenpairs of a symmetric / Hermitian operator.
    src/IterativeSolvers/EIGHS/eighs.fypp:46-140.  Returns (eigvals[nev], residuals[nev], info = Lanczos steps).
    (`pipelined`, engine extra as in eigs: the Lanczos steps are enqueued in asynchronous device segments while the host
    runs the per-step `eigh` tests of the previous segment on several threads; same results.)"""
    nev = len(X)
    kdim_ = 4 * nev if kdim is None else kdim
    proto = X[0]
    dt = _dtype_of(proto)
    Xwrk = _new_basis(proto, kdim_ + 1)                                            # allocate(Xwrk, mold=X(1)); zero_basis   :60
    if x0 is not None:
        copy(Xwrk[0], x0)
        Xwrk[0].scal(1.0 / x0.norm())
    else:
        Xwrk[0].rand(True)
    T = np.zeros((kdim_ + 1, kdim_), dtype=dt, order="F")
    vals = np.zeros(kdim_)
    vecs = np.zeros((kdim_, kdim_), dtype=dt)
    res = np.zeros(kdim_)

    def ritz_test(k):
        w, v = _hostlapack.syev(T[:k, :k])                                         # :87  (stdlib eigh = syev / heev: ascending)
        return w, v, np.abs(T[k, k - 1] * v[k - 1, :k])                            # :93

    def accept(k, w, v, r) -> bool:
        vals[:] = 0                                                                # :86
        vecs[:] = 0
        vals[:k], vecs[:k, :k], res[:k] = w, v, r
        if write_intermediate:
            write_results("eighs_output.txt", vals[:k].astype(complex), res[:k], tolerance)
        return np.count_nonzero(res[:k] < tolerance) >= nev                        # :96, :101

    can_pipeline = (isinstance(Xwrk, krylov_basis_gpu) and isinstance(A, _engine_linop) and not write_intermediate
                    and kdim_ <= 512)
    pipelined = can_pipeline and (_hostlapack.threaded() if pipelined is None else bool(pipelined))
    nthreads = max(1, min(32, os.cpu_count() or 1))
    k, k_from, stopped = 0, 1, False
    if pipelined:
        bounds = [(a, min(a + _EIGS_SEGMENT - 1, kdim_)) for a in range(1, kdim_ + 1, _EIGS_SEGMENT)]
        pool, device = _pool("geev", nthreads), _pool("device", 1)
        fut = device.submit(lanczos, A, Xwrk, T, bounds[0][0], bounds[0][1])
        klast = 0
        for si, (a, b) in enumerate(bounds):
            linfo = fut.result()
            fut = None
            klast = linfo if linfo > 0 else b                                      # a breakdown ends the batch: step by step from there
            if linfo == 0 and si + 1 < len(bounds):
                fut = device.submit(lanczos, A, Xwrk, T, bounds[si + 1][0], bounds[si + 1][1])
            for c0 in range(a, klast + 1, nthreads):
                ks = range(c0, min(c0 + nthreads, klast + 1))
                for k, (w, v, r) in zip(ks, pool.map(ritz_test, ks)):
                    if accept(k, w, v, r):
                        stopped = True
                        break
                if stopped:
                    break
            if stopped or linfo > 0:
                break
        if fut is not None:
            fut.result()                                                           # a segment in flight beyond the stop: only T(:, k+1:) and Xwrk(k+2:) changed
        k_from = klast + 1
    if not stopped:
        for k in range(k_from, kdim_ + 1):
            lanczos(A, Xwrk, T, kstart=k, kend=k)                                  # :84
            if accept(k, *ritz_test(k)):
                break
    idx = np.argsort(-vals, kind="stable")                                         # sort_index(..., reverse=.true.) over all kdim   :106
    vals, vecs, res = vals[idx], vecs[:, idx], res[idx]
    k = min(k, kdim_)
    Y = linear_combination(Xwrk[:k], np.asfortranarray(vecs[:k, :nev].astype(dt)))  # X(i) = sum_j eigvecs(j, i) Xwrk(j)   :113-118
    copy(X, Y)
    return vals[:nev].copy(), res[:nev].copy(), k


@_hostlapack.small_problems
def svds(A: abstract_linop, U, V, u0: abstract_vector | None = None, kdim: int | None = None,
         tolerance: float = rtol_dp, write_intermediate: bool = False, pipelined: bool | None = None):
    """Golub-Kahan solver for the leading len(U) singular triplets.  src/IterativeSolvers/SVDS/svd_solvers.fypp:44-150.
    Returns (S[nsv], residuals[nsv], info).  (`pipelined`: engine extra as in eigs / eighs, same results.)"""
    nsv = len(U)
    kdim_ = 4 * nsv if kdim is None else kdim
    dt = _dtype_of(U[0])
    Uwrk = _new_basis(U[0], kdim_ + 1)
    if u0 is not None:
        copy(Uwrk[0], u0)
        Uwrk[0].scal(1.0 / u0.norm())
    else:
        Uwrk[0].rand(True)
    Vwrk = _new_basis(V[0], kdim_ + 1)
    B = np.zeros((kdim_ + 1, kdim_), dtype=dt, order="F")
    sv = np.zeros(kdim_)
    um = np.zeros((kdim_, kdim_), dtype=dt)
    vm = np.zeros((kdim_, kdim_), dtype=dt)
    res = np.zeros(kdim_)

    def ritz_test(k):
        u, s_, vh = _hostlapack.gesdd(B[:k, :k])                                   # :102  (stdlib svd = gesdd)
        v = vh.conj().T                                                            # vmat = hermitian(vmat)   :104
        return u, s_, v, np.abs(B[k, k - 1] * v[k - 1, :k])                        # :106

    def accept(k, u, s_, v, r) -> bool:
        sv[:] = 0
        um[:] = 0
        vm[:] = 0
        sv[:k], um[:k, :k], vm[:k, :k], res[:k] = s_, u, v, r
        if write_intermediate:
            write_results("svds_output.txt", sv[:k].astype(complex), res[:k], tolerance)
        return np.count_nonzero(res[:k] < tolerance) >= nsv

    can_pipeline = (isinstance(Uwrk, krylov_basis_gpu) and isinstance(Vwrk, krylov_basis_gpu) and isinstance(A, _engine_linop)
                    and not write_intermediate and kdim_ <= 512 and tolerance >= atol_dp)
    pipelined = can_pipeline and (_hostlapack.threaded() if pipelined is None else bool(pipelined))
    nthreads = max(1, min(32, os.cpu_count() or 1))
    k, k_from, stopped = 0, 1, False
    if pipelined:
        bounds = [(a, min(a + _EIGS_SEGMENT - 1, kdim_)) for a in range(1, kdim_ + 1, _EIGS_SEGMENT)]
        pool, device = _pool("geev", nthreads), _pool("device", 1)
        fut = device.submit(bidiagonalization, A, Uwrk, Vwrk, B, bounds[0][0], bounds[0][1], tolerance)
        klast = 0
        for si, (a, b) in enumerate(bounds):
            binfo = fut.result()
            fut = None
            klast = binfo if binfo > 0 else b                                      # a breakdown ends the batch: step by step from there
            if binfo == 0 and si + 1 < len(bounds):
                fut = device.submit(bidiagonalization, A, Uwrk, Vwrk, B, bounds[si + 1][0], bounds[si + 1][1], tolerance)
            for c0 in range(a, klast + 1, nthreads):
                ks = range(c0, min(c0 + nthreads, klast + 1))
                for k, out in zip(ks, pool.map(ritz_test, ks)):
                    if accept(k, *out):
                        stopped = True
                        break
                if stopped:
                    break
            if stopped or binfo > 0:
                break
        if fut is not None:
            fut.result()                                                           # a segment in flight beyond the stop: later columns only
        k_from = klast + 1
    if not stopped:
        for k in range(k_from, kdim_ + 1):
            bidiagonalization(A, Uwrk, Vwrk, B, kstart=k, kend=k, tol=tolerance)  # :96
            if accept(k, *ritz_test(k)):
                break
    k = min(k, kdim_)
    copy(U, linear_combination(Uwrk[:k], np.asfortranarray(um[:k, :nsv].astype(dt))))   # :121-127
    copy(V, linear_combination(Vwrk[:k], np.asfortranarray(vm[:k, :nsv].astype(dt))))
    return sv[:nsv].copy(), res[:nsv].copy(), k
