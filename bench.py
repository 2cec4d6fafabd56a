#!/usr/bin/env python3
"""bench.py -- Arnoldi-iteration throughput + DGS sweep bandwidth on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 without a launcher's environment it starts
                                                            `python -m torch.distributed.run --nproc-per-node N bench.py ...`
                                                            as a child process, relays its output and exits with its code)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form for N > 1)

A "step" is ONE Arnoldi factorisation of the workload (m Arnoldi iterations: operator kernel,
three fused DGS panel sweeps, normalise), run through the C ABI (lk_arnoldi).  Workload at every N:
BASELINE.json's metric configuration -- synthetic diagonal operator d_i = 1 + i/n, n = 10^8
real(dp) rows, m = 128, x0 from the shared counter RNG -- row-sharded over the N ranks (strong
scaling: total work fixed), the only cross-rank traffic being the RCCL all-reduce of the <= 129
reduction scalars after each sweep.  Inputs are generated in HBM before the timed region.

`--operator dense | lap5 | csr` run the same factorisation on the other synthetic operators north_star names (dense matvec,
5-point stencil, the same Laplacian as a CSR matrix; defaults: n = 65536 / 4096^2 / 4096^2), each row-sharded over the N ranks
(dense, CSR: row block of A + all-gather of x; stencil: one grid line to each neighbour).  The headline line is the default
(diagonal) operator.

Every phase writes one marker line per rank to stderr and re-arms a watchdog (`--watchdog` seconds, 600): a rank that stops moving dumps
every thread's stack and exits 1 (class Progress); the reduction route (native RCCL or a torch.distributed callback) is agreed on by all
ranks; `--gpus N` under a launcher with another WORLD_SIZE is refused.  `--dump-h PATH`: rank 0 saves the last Hessenberg matrix.

Prints ONE JSON line on rank 0; `value` = Arnoldi iterations per second, whole job.
  roofline   : the DGS sweep kernels (lk::panel_dot_cw for sweep 1, lk::panel_sweep for sweeps 2 and 3) -- algorithmic bytes s*n_local*(k+1|k+2)
               per launch (s*n*(3k+5) per DGS) / HIP-event duration on the kernel's
               stream, averaged over every sweep launch of the timed region; peak = 8 TB/s HBM3E.
  cpu_baseline: the reference-schedule CPU oracle (oracle/, 1 thread, kind "port") timed on this
               host on a bounded sample and scaled to the metric's unit (see `sample`), the byte model
               checked on a second sample; plus an all-core leg running the engine's fused schedule.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# Reduction order of the N > 1 all-reduce: RCCL picks ring / tree (and the protocol) per call from the message size and the
# topology it detects; pinning the algorithm makes the summation order of the <= 129 scalars a function of (N, rank order) only,
# so a run is bit-reproducible from launch to launch.  A caller's own setting wins.
NCCL_PIN = {"NCCL_ALGO": "Ring"}


def _self_launch(ngpus: int) -> int:
    """--gpus N > 1 without a launcher: run this script under torch.distributed.run in a FRESH child process (decided before
    anything in this process touches the GPU; never an exec), relay its output, return its exit code.  `--standalone` lets the
    launcher's own store pick and HOLD a free port (a port probed here and released could be taken before the launcher binds
    it); `--local-addr 127.0.0.1` keeps the rendezvous on loopback (the box's hostname may not resolve)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k, v in NCCL_PIN.items():
        env.setdefault(k, v)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, cwd=ROOT)
    return proc.returncode


class Progress:
    """Where is this rank, and is it still moving?  Every phase of a run writes ONE marker line to stderr
    (`bench.py[rank r/N pid P] +12.3s phase: ...`) and re-arms a watchdog: `seconds` without the next marker (or `tick`) and
    the interpreter's fault handler -- a C thread that needs neither the interpreter lock nor a responsive main thread --
    dumps the Python stack of EVERY thread to stderr and ends the process with a fresh `_exit(1)` (never a re-exec of a
    process that has touched the GPU).  Under torch.distributed.run one rank exiting non-zero makes the launcher terminate
    the others, whose SIGTERM handler dumps their stacks too: a hang becomes rc != 0 plus the place every rank was stuck at,
    within `seconds`.  SIGUSR1 dumps the stacks without ending the process."""

    def __init__(self, seconds: float, rank: int = 0, world: int = 1):
        import faulthandler
        import signal
        self.fh, self.seconds, self.rank, self.world, self.t0 = faulthandler, float(seconds), rank, world, time.perf_counter()
        self.last = "start"
        faulthandler.enable(all_threads=True)
        try:
            faulthandler.register(signal.SIGUSR1, all_threads=True, chain=False)
            faulthandler.register(signal.SIGTERM, all_threads=True, chain=True)
        except (AttributeError, ValueError, RuntimeError):      # not the main thread / no such signal: markers still work
            pass

    def _arm(self) -> None:
        if self.seconds > 0:
            self.fh.dump_traceback_later(self.seconds, repeat=False, file=sys.stderr, exit=True)

    def phase(self, name: str) -> None:
        self.last = name
        print(f"bench.py[rank {self.rank}/{self.world} pid {os.getpid()}] +{time.perf_counter() - self.t0:.1f}s phase: {name}"
              + (f" (watchdog {self.seconds:.0f}s)" if self.seconds > 0 else ""), file=sys.stderr, flush=True)
        self._arm()

    def tick(self) -> None:
        """progress inside a phase (one factorisation done): re-arm without a marker line"""
        self._arm()

    def done(self) -> None:
        self.fh.cancel_dump_traceback_later()


def _die(progress: "Progress", msg: str, code: int = 3) -> None:
    """A failure the other ranks cannot be told about (they may be inside a collective this rank never entered): say so and
    leave with a non-zero status at once -- the launcher then terminates the whole job.  `os._exit`: no destructor of this
    process may wait for a peer on the way out."""
    print(f"bench.py[rank {progress.rank}/{progress.world} pid {os.getpid()}] FATAL after phase '{progress.last}': {msg}",
          file=sys.stderr, flush=True)
    sys.stdout.flush()
    os._exit(code)


def _install_faulty_reduction(ctx, torch, factor: float) -> None:
    """Test hook (LK_BENCH_TEST_HOOKS=1 + LK_TEST_SCALE_PARTIALS_RANK): wrap the all-reduce callback the context installed with one
    that first scales this rank's partial sums in device memory -- through the public ABI (lk_set_allreduce), nothing inside the library."""
    from lightkrylov_amd import _capi
    from lightkrylov_amd.context import _DevMem
    inner = ctx._cb

    def faulty(user, dev_ptr, count, stream_ptr):
        st = torch.cuda.ExternalStream(int(stream_ptr), device=ctx.device) if stream_ptr else torch.cuda.current_stream(ctx.device)
        with torch.cuda.stream(st):
            torch.as_tensor(_DevMem(int(dev_ptr), int(count)), device=f"cuda:{ctx.device}").mul_(factor)
        return inner(user, dev_ptr, count, stream_ptr)

    ctx._cb_faulty = _capi.ALLREDUCE_FN(faulty)
    _capi.check(ctx._lib.lk_set_allreduce(ctx._h, ctx._cb_faulty, None, ctx.nranks, ctx.rank))


def load_traffic_record(n_local: int, m: int, dtype: str):
    """The stored PMC traffic record for a sweep workload (n_local rows per rank, m, dtype), or (None, why).  profiles/pmc_traffic.json
    holds one record per n_local -- the sweep kernels of a rank see only their row block, so ONE GPU measures the record of every shard size
    (bench.py --shard-of P under rocprofv3 --pmc).  A record names the SHA-256 of the kernel sources it was measured on; a record made on
    other kernels is refused loudly instead of being quoted as if it described this build."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc):
        return None, "rocprofv3 --pmc passes cannot run inside bench.py; see profiles/ (no pmc_traffic.json)"
    try:
        doc = json.load(open(pmc))
        recs = doc.get("records", [doc]) if isinstance(doc, dict) else list(doc)
        rec = next((r for r in recs if r.get("n_local") == n_local and r.get("m") == m and r.get("dtype") == dtype), None)
        if rec is None:
            return None, (f"profiles/pmc_traffic.json has no record for n_local={n_local}, m={m}, {dtype} "
                          f"(has n_local = {sorted(r.get('n_local') for r in recs)})")
        have = kernel_source_hash()
        if rec.get("kernel_source_sha256") != have:
            why = (f"STALE: profiles/pmc_traffic.json (n_local={n_local}) was measured on kernel sources {str(rec.get('kernel_source_sha256'))[:12]}..., "
                   f"this build is {have[:12]}...; re-run tools/run_profiles.sh")
            print("bench.py: " + why, file=sys.stderr)
            return None, why
        return rec.get("hbm_bytes_per_launch"), {k: rec.get(k) for k in ("commit", "kernel_source_sha256", "source", "method", "traffic_over_algorithmic") if k in rec}
    except Exception as exc:  # noqa: BLE001
        return None, f"profiles/pmc_traffic.json unreadable: {exc!r}"


def predicted_iters_per_s(n: int, m: int, world: int, operator: str, dtype: str):
    """DESIGN.md section 6's scaling MODEL for the metric workload, printed next to the measured `value` so that a scaling run explains its
    own gap: a factorisation costs T1 / N (the sweeps, the operator and the normalise are rank-local and bandwidth-bound; T1 = the measured
    single-GPU factorisation, profiles/scaling_model.json) plus what does not shrink -- per Arnoldi step three all-reduces of <= 129 doubles
    and three finish kernels.  Returns None when the repository holds no single-GPU record for this workload."""
    path = os.path.join(ROOT, "profiles", "scaling_model.json")
    if not os.path.exists(path):
        return None
    try:
        mod = json.load(open(path))
        w = mod["workload"]
        if (w["operator"], w["dtype"], w["n_global"], w["m"]) != (operator, dtype, n, m):
            return None
        t1 = mod["single_gpu_ms_per_factorisation"] * 1e-3
        fixed = 0.0 if world == 1 else m * 3 * (mod["allreduce_us_assumed"] + mod["finish_partials_us"]) * 1e-6
        shard = mod.get("shard_ms_per_factorisation", {}).get(str(world))
        local = shard * 1e-3 if shard else t1 / world
        t = local + fixed
        return {"predicted_it_s": m / t, "predicted_ms_per_step": 1e3 * t,
                "model": (f"local work of a rank ({'measured on one GPU: bench.py --shard-of ' + str(world) if shard else 'T1 / N = ' + format(1e3 * t1, '.1f') + ' ms / ' + str(world)}"
                          f" = {1e3 * local:.1f} ms) + m*3*(allreduce + finish) = {m}*3*({mod['allreduce_us_assumed']} + {mod['finish_partials_us']}) us"),
                "single_gpu_record": mod.get("source"), "note": "a model from single-GPU measurements (DESIGN.md section 6), not a measurement"}
    except Exception as exc:  # noqa: BLE001
        return {"predicted_it_s": None, "note": f"profiles/scaling_model.json unreadable: {exc!r}"}


def launch_bound_regime(ctx, n: int = 175_000, m: int = 64, reps: int = 6) -> dict:
    """Context line (not the metric): the reference's own published problem size (1.75e5 unknowns, paper/paper.md:103-113), real(dp),
    m = 64, diagonal operator -- Arnoldi iterations/s of lk_arnoldi with the single-launch Gram-Schmidt step (csrc/lk_resident.hip.h)
    and on the three-sweep schedule, interleaved, best of `reps`; run AFTER the timed region, a few tens of milliseconds."""
    import lightkrylov_amd as lk
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=ctx)
    H = np.zeros((m + 1, m), order="F")
    best = {0: float("inf"), 1: float("inf")}
    before = ctx.resident_stats()
    try:
        for rep in range(reps + 1):
            for route in (0, 1):
                ctx.set_tuning("resident", route)
                X[0].rand(True, seed=7)
                ctx.sync()
                t0 = time.perf_counter()
                info = lk.arnoldi(A, X, H)
                dt = time.perf_counter() - t0
                if info != 0:
                    raise RuntimeError(f"arnoldi info = {info}")
                if rep:
                    best[route] = min(best[route], dt)
    finally:
        ctx.set_tuning("resident", 1)
    after = ctx.resident_stats()
    return {"workload": f"arnoldi n={n} m={m} f64 diag (the reference's published problem size), one GPU, after the timed region",
            "unit": "Arnoldi iterations/s", "single_launch_step": round(m / best[1], 1), "three_sweeps": round(m / best[0], 1),
            "single_launches": int(after[0] - before[0]), "gave_up": int(after[1] - before[1]), "register_resident": int(after[2] - before[2])}


def kernel_source_hash() -> str:
    """SHA-256 over the device code and its launcher: what a PMC traffic record must have been measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("lk_kernels.hip.h", "lk_engine.hip", "lk_resident.hip.h"):
        h.update(open(os.path.join(ROOT, "lightkrylov_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def _oracle_sample(n: int, m: int, threads: int, fused: bool, repeat: int = 1):
    """Timed oracle Arnoldi on the bench workload's formulas (diag-linspace operator, counter-RNG x0); the last of
    `repeat` runs is the one timed."""
    from oracle import oracle as ora
    ora.set_threads(threads)
    X = np.zeros((n, m + 1), order="F")
    H = np.zeros((m + 1, m), order="F")
    A = ora.DiagLinOp(1.0, 1.0 / n)
    dt, info = 0.0, 0
    for _ in range(repeat):
        ora.fill_counter(X[:, 0], 7)
        ora.scal(X[:, 0], 1.0 / ora.norm(X[:, 0]))
        t0 = time.perf_counter()
        info = ora.arnoldi_fused_allcores(A, X, H) if fused else ora.arnoldi(A, X, H)
        dt = time.perf_counter() - t0
    ora.set_threads(1)
    return dt, int(info)


def _cpu_quota_cores():
    """CPUs' worth of time this process's cgroup may use (cpu.max of cgroup v2, cfs quota of v1), or None when unlimited / unknown: a container
    that SEES every hardware thread of the host may still be allowed only a few of them -- threads beyond the quota are throttled, not run."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            return float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def _fused_leg(ns: int, ms: int, n4: int, m4: int, n_full: int, m_full: int) -> dict:
    """The all-core leg itself (runs in the child process): thread scan on (ns, ms), then the timed sample (n4, m4) on the best count."""
    from oracle import oracle as ora
    fus_model = lambda nn, mm: sum(8.0 * nn * (3 * k + 5 + 2 + 3) for k in range(1, mm + 1))  # noqa: E731

    def sample(n, m, T, repeat):
        ora.set_threads(T)
        X = np.zeros((n, m + 1), order="F")
        ora.first_touch(X)                                   # pages placed by the threads that will stream them
        H = np.zeros((m + 1, m), order="F")
        A = ora.DiagLinOp(1.0, 1.0 / n)
        dt = 0.0
        for _ in range(repeat):
            ora.fill_counter(X[:, 0], 7)
            ora.scal(X[:, 0], 1.0 / ora.norm(X[:, 0]))
            t0 = time.perf_counter()
            ora.arnoldi_fused_allcores(A, X, H)
            dt = time.perf_counter() - t0
        return dt

    tmax = ora.max_threads()
    quota = _cpu_quota_cores()
    scan = {}
    # (a team more than four times the cgroup's CPU quota is not tried: every thread beyond the quota only adds throttling -- 256 threads
    # on a 16-CPU grant took 20 s for the 0.6 s sample)
    tcap = tmax if quota is None else min(tmax, max(8, int(4 * quota)))
    for T in sorted({t for t in (8, 16, 32, 64, 96, 128, 192, 256, tmax) if t <= tcap}):
        scan[T] = sample(ns, ms, T, 2)
    T = min(scan, key=scan.get)
    ora.set_threads(T)
    cpus = ora.thread_cpus()
    dt4 = sample(n4, m4, T, 2)
    bw4 = fus_model(n4, m4) / dt4
    return {
        "value": m_full / (fus_model(n_full, m_full) / bw4), "unit": "Arnoldi iterations/s", "cores": T,
        "sample_seconds": dt4, "sample_iters_per_s": m4 / dt4, "effective_GBps_on_fused_schedule": bw4 / 1e9,
        "thread_scan_seconds": {str(k): v for k, v in scan.items()},
        "thread_scan_GBps": {str(k): fus_model(ns, ms) / v / 1e9 for k, v in scan.items()},
        "host_threads_available": tmax,
        "cgroup_cpu_quota_cores": quota,
        "why_not_more_threads": (f"the job's cgroup grants {quota:.0f} CPUs' worth of time on a host that shows {tmax} hardware threads: a team larger than that is "
                                 "throttled by the scheduler, not run (the scan above shows it)") if quota is not None and quota < tmax else None,
        "thread_scan_sample": {"n": ns, "m": ms, "basis_GB": 8e-9 * ns * (ms + 1)},
        "binding": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"),
                    "distinct_cpus_of_the_team": len(set(cpus)), "first_touch": "parallel, by the team that streams the rows"},
        "sample": f"three-sweep fused CGS2 (the engine's schedule) with OpenMP on {T} bound threads (best of the scan), n={n4}, m={m4} "
                  f"({dt4:.2f} s) scaled by sum_k 8n(3k+10) bytes",
    }


def _fused_leg_in_child(budget_n: int, budget_m: int, n_full: int, m_full: int, progress=None) -> dict:
    import subprocess
    ns, ms = budget_n, max(budget_m + budget_m // 2, 4) + 2           # the scan's own sample is DRAM-resident: >= 2 GB of basis
    n4, m4 = min(4 * budget_n, n_full), min(2 * budget_m, m_full)
    env = dict(os.environ, OMP_PROC_BIND="spread", OMP_PLACES="cores")
    env.pop("OMP_NUM_THREADS", None)
    spec = json.dumps([ns, ms, n4, m4, n_full, m_full])
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-fused-leg", spec], env=env, cwd=ROOT, capture_output=True,
                             text=True, timeout=max(60.0, (progress.seconds - 30.0) if progress is not None and progress.seconds > 0 else 900.0))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode != 0 or not line:
            return {"value": None, "unit": "Arnoldi iterations/s", "cores": None, "sample": f"child failed (rc {out.returncode}): {out.stderr[-400:]}"}
        return json.loads(line[-1])
    except Exception as exc:  # noqa: BLE001
        return {"value": None, "unit": "Arnoldi iterations/s", "cores": None, "sample": f"child failed: {exc!r}"}


def cpu_baseline(n_full: int, m_full: int, budget_n: int, budget_m: int, progress=None) -> dict:
    """Two legs, both on this host, both on a BOUNDED sample scaled to the metric's unit by a byte model:
      reference_schedule_1thread -- the oracle's Arnoldi in the reference's own schedule (per-primitive BLAS-1,
        sequential dots, scal-then-axpy axpby, fresh projection vector per pass; ONE thread: the reference has no
        threading, src/Krylov/gram_schmidt.fypp has no parallel region).  Byte model (14k+20)*s*n per DGS + 3*s*n
        for the diagonal matvec.  A second, differently shaped sample checks the model.
      fused_allcores -- the engine's three-sweep fused schedule on every host core (OpenMP): the best the host can
        do with the same algorithm, byte model (3k+5)*s*n + normalise 2 + matvec 3.
    `value` (the headline of this object) is the reference-schedule leg: that is what a LightKrylov user runs."""
    from oracle import oracle as ora
    ref_model = lambda nn, mm: sum(8.0 * nn * (14 * k + 20 + 3) for k in range(1, mm + 1))   # noqa: E731
    fus_model = lambda nn, mm: sum(8.0 * nn * (3 * k + 5 + 2 + 3) for k in range(1, mm + 1))  # noqa: E731
    # -- leg 1: reference schedule, 1 thread.  Three DRAM-resident samples: two fit t = n (a m + b m(m+1)/2)
    #    (a: per-step cost independent of k -- matvec, norms, allocations; b: cost per basis column), the
    #    third validates the fit; the byte model alone (one sample) is reported beside it.
    #    The second sample carries m >= 44, so the fit is extrapolated by < 3x in m to the metric's m = 128 (and linearly in n,
    #    which a DRAM-bound stream is).
    n1, m1 = budget_n, max(budget_m // 2, 4)
    n2, m2 = max(budget_n // 4, 1000), max(min(2 * budget_m + 4, m_full), 4)
    n3, m3 = max(budget_n // 2, 1000), max(budget_m + budget_m // 5, 4)
    dt1, info1 = _oracle_sample(n1, m1, 1, fused=False)
    if progress is not None:
        progress.tick()
    dt2, _ = _oracle_sample(n2, m2, 1, fused=False)
    if progress is not None:
        progress.tick()
    dt3, _ = _oracle_sample(n3, m3, 1, fused=False)
    if progress is not None:
        progress.tick()
    tri = lambda mm: mm * (mm + 1) / 2.0                                                      # noqa: E731
    Afit = np.array([[n1 * m1, n1 * tri(m1)], [n2 * m2, n2 * tri(m2)]], dtype=float)
    a, b = np.linalg.solve(Afit, np.array([dt1, dt2]))
    pred3 = n3 * (a * m3 + b * tri(m3))
    fit_ok = a >= 0 and b > 0
    bw3 = ref_model(n3, m3) / dt3
    t_full_bytes = ref_model(n_full, m_full) / bw3
    t_full_fit = n_full * (a * m_full + b * tri(m_full))
    t_full = t_full_fit if fit_ok else t_full_bytes
    leg1 = {
        "value": m_full / t_full, "unit": "Arnoldi iterations/s", "cores": 1,
        "samples": [{"n": n1, "m": m1, "seconds": dt1}, {"n": n2, "m": m2, "seconds": dt2},
                    {"n": n3, "m": m3, "seconds": dt3, "predicted_seconds_by_fit": pred3,
                     "model_error": abs(pred3 - dt3) / dt3, "within_10_percent": bool(abs(pred3 - dt3) / dt3 <= 0.10)}],
        "fit_seconds_per_row": {"per_step": a, "per_step_per_column": b, "used": bool(fit_ok)},
        "extrapolation": {"in_m": m_full / max(m1, m2), "in_n": n_full / max(n1, n2)},
        "value_by_byte_model_only": m_full / t_full_bytes,
        "effective_GBps_on_reference_schedule": bw3 / 1e9,
        "sample": (f"samples (n, m) = ({n1}, {m1}), ({n2}, {m2}) fit t = n(a m + b m(m+1)/2); validated on ({n3}, {m3}): "
                   f"{dt3:.2f} s measured vs {pred3:.2f} s predicted"),
    }
    # -- leg 2: fused schedule on the host cores, in a CHILD process started with OMP_PROC_BIND=spread / OMP_PLACES=cores (thread binding is
    #    read when the OpenMP runtime starts -- this process loaded it long ago with torch) so that the threads stay where they first touched
    #    their rows of the basis: on a two-socket host an unbound team reads most of the basis across the socket link (round 4: best at 16 of
    #    128 threads, 152 GB/s).  The child never touches the GPU.  More threads is still not monotonically faster, so a short scan picks the
    #    thread count, which is what a user tuning OMP_NUM_THREADS would do; the scan is reported.
    if progress is not None:
        progress.phase("cpu_baseline: all-core leg (fused schedule, bound threads, NUMA first touch) in a child process")
    leg2 = _fused_leg_in_child(budget_n, budget_m, n_full, m_full, progress)
    return {
        "value": leg1["value"], "unit": "Arnoldi iterations/s", "cores": 1, "kind": "port", "info": info1,
        "host_cpus": os.cpu_count(),
        "sample": ("oracle (C restatement of the reference schedule, 1 thread) arnoldi, diagonal operator, real(dp): "
                   + leg1["sample"] + f"; scaled to n={n_full}, m={m_full}"),
        "reference_schedule_1thread": leg1,
        "fused_allcores": leg2,
    }


def cpu_baseline_operator(operator: str, n_full: int, m_full: int) -> dict:
    """cpu_baseline of the operator legs: the oracle's Arnoldi in the reference's schedule, ONE thread, on a bounded sample of
    the same operator family, scaled to the full workload by a byte model (matvec bytes + (14k+20)*s*n per DGS)."""
    from oracle import oracle as ora
    ora.set_threads(1)
    dgs = lambda nn, mm: sum(8.0 * nn * (14 * k + 20) for k in range(1, mm + 1))   # noqa: E731
    if operator == "dense":
        ns, ms = 8192, 12
        rng = np.random.default_rng(0)
        op = ora.DenseOp(np.asfortranarray(rng.uniform(-1.0, 1.0, (ns, ns))))
        mvb = lambda nn: 8.0 * nn * nn                                                # noqa: E731
        what = f"dense {ns} x {ns} (the stdlib gemv loop of the reference, AbstractLinops.fypp:623)"
    else:
        Ns = 1448                                                                     # ~2.1e6 rows
        ns, ms = Ns * Ns, 16
        op = ora.Lap5Op(Ns)
        mvb = lambda nn: 16.0 * nn                                                    # noqa: E731
        what = f"5-point Laplacian {Ns} x {Ns} (the oracle's stencil loop; the reference has no sparse type of its own)"
    X = np.zeros((ns, ms + 1), order="F")
    H = np.zeros((ms + 1, ms), order="F")
    ora.fill_counter(X[:, 0], 7)
    ora.scal(X[:, 0], 1.0 / ora.norm(X[:, 0]))
    t0 = time.perf_counter()
    info = ora.arnoldi(op, X, H)
    dt = time.perf_counter() - t0
    bw = (ms * mvb(ns) + dgs(ns, ms)) / dt
    t_full = (m_full * mvb(n_full) + dgs(n_full, m_full)) / bw
    return {"value": m_full / t_full, "unit": "Arnoldi iterations/s", "cores": 1, "kind": "port", "info": int(info), "host_cpus": os.cpu_count(),
            "sample_seconds": dt, "effective_GBps_on_reference_schedule": bw / 1e9,
            "sample": f"oracle (C restatement of the reference schedule, 1 thread) arnoldi, {what}, m={ms}: {dt:.2f} s; scaled to n={n_full}, "
                      f"m={m_full} by bytes (operator + (14k+20)*8*n per DGS)"}


def _laplacian_csr_rows(N: int, row0: int, n_local: int):
    """Rows [row0, row0 + n_local) of the 5-point Laplacian on an N x N grid (Dirichlet, scaled by (N+1)^2) as
    (rowptr, colind, vals) with global column indices in ascending order per row -- the stencil operator's matrix."""
    i = np.arange(row0, row0 + n_local, dtype=np.int64)
    ix, jy = i % N, i // N
    sc = float((N + 1) ** 2)
    cols = np.stack([i - N, i - 1, i, i + 1, i + N], axis=1)
    ok = np.stack([jy > 0, ix > 0, np.ones_like(i, dtype=bool), ix < N - 1, jy < N - 1], axis=1)
    vals = np.broadcast_to(np.array([-sc, -sc, 4.0 * sc, -sc, -sc]), cols.shape)
    rowptr = np.zeros(n_local + 1, dtype=np.int64)
    np.cumsum(ok.sum(axis=1), out=rowptr[1:])
    return rowptr, cols[ok].astype(np.int32), np.ascontiguousarray(vals[ok])


def main() -> None:
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-fused-leg":      # the child of cpu_baseline's all-core leg: host only, no torch, no GPU
        print(json.dumps(_fused_leg(*json.loads(sys.argv[2]))), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: the launcher's WORLD_SIZE when there is one, else 1); an EXPLICIT value that "
                         "disagrees with a launcher's WORLD_SIZE is refused")
    ap.add_argument("--shard-of", type=int, default=0, metavar="P",
                    help="diagnostic, single process: run ONE row block of a P-rank job (n_local = rows / P, partition announced as "
                         "rows of the global problem, no communicator) -- what the sweep kernels of a rank see; used to take the "
                         "per-shard PMC traffic records on one GPU (tools/run_profiles.sh).  Not a metric line.")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--operator", default="diag", choices=["diag", "dense", "lap5", "csr"],
                    help="synthetic operator: diag (the metric's configuration), dense (n x n matrix: GEMV), lap5 (5-point stencil), "
                         "csr (the same Laplacian as an explicit sparse matrix)")
    ap.add_argument("--rows", dest="n", type=int, default=None,
                    help="global rows (default: 1e8 for diag = the metric config; 65536 for dense; 4096^2 for lap5 / csr)")
    ap.add_argument("--kdim", dest="m", type=int, default=128, help="Krylov dimension (metric config: 128)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "c128"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=8_000_000)
    ap.add_argument("--cpu-m", type=int, default=20)
    ap.add_argument("--grid-mult", type=int, default=0)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT", help="lk_set_tuning knob (repeatable)")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("LK_BENCH_WATCHDOG", "600")),
                    help="seconds without progress after which every thread's stack is dumped to stderr and the process exits 1 "
                         "(0 = off; env LK_BENCH_WATCHDOG)")
    ap.add_argument("--dump-h", default=None, metavar="PATH", help="rank 0 saves the Hessenberg matrix of the last factorisation (.npy): parity checks of sharded runs")
    ap.add_argument("--no-profile", action="store_true",
                    help="diagnostic: leave the library's per-kernel HIP events off (value only; roofline fields are then zero)")
    args = ap.parse_args()
    parity_failed = False

    under_launcher = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    gpus_given = args.gpus is not None
    if not gpus_given:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1     # the launcher's size wins when --gpus is silent
    if args.gpus > 1 and not under_launcher:
        sys.exit(_self_launch(args.gpus))
    # test hooks (tests/test_gpu_distributed.py) are honoured only with the explicit opt-in LK_BENCH_TEST_HOOKS=1: a leaked LK_TEST_*
    # variable alone never changes a metric run
    hooks = os.environ.get("LK_BENCH_TEST_HOOKS") == "1"
    hook = lambda name: os.environ.get(name) if hooks else None                                 # noqa: E731
    for k, v in NCCL_PIN.items():
        os.environ.setdefault(k, v)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL between processes needs on this driver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    progress = Progress(args.watchdog, rank, world)
    if gpus_given and world != args.gpus:
        # a launcher's environment that disagrees with an EXPLICIT --gpus (a stale WORLD_SIZE export, a scheduler's variables):
        # running on the launcher's size would hand the caller a number for a job it did not ask for -- refuse, on every rank
        print(f"bench.py[rank {rank}/{world}]: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run", file=sys.stderr, flush=True)
        sys.exit(2)
    progress.phase("importing torch and the engine")

    import torch
    import lightkrylov_amd as lk

    # test hooks (tests/test_gpu_distributed.py): several ranks on ONE device with the gloo backend, which
    # all-reduces CUDA tensors through the host -- exercises the multi-process sharded path where RCCL cannot
    # (RCCL refuses two ranks on one GPU).  Never set by the driver.
    backend = os.environ.get("LK_DIST_BACKEND", "nccl")
    if "LK_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["LK_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run: RCCL path even for one rank
        import torch.distributed as dist  # noqa: PLW0621
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        progress.phase(f"rendezvous: init_process_group({backend}) on device {local_rank}, store {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
        progress.phase("rendezvous done; first collective (barrier)")
        dist.barrier()

    progress.phase("creating the engine context")
    if hook("LK_TEST_HANG_RANK") == str(rank):                            # test hook: this rank stops making progress
        time.sleep(1e6)
    ctx = lk.Context(device=local_rank)
    reduce_path = "none (single rank)"
    if dist is not None:
        # data-path collective: ncclAllReduce issued by the library itself on its own stream (lk_comm_init_rank);
        # torch.distributed only ships the 128 bootstrap bytes and provides barrier / max-over-ranks for timing.
        # LK_NATIVE_RCCL=0 selects the older route (torch.distributed.all_reduce through a ctypes callback), `force` tries the
        # native one whatever the torch backend (test hook).  The route is AGREED ON by all ranks -- the wish (an environment
        # variable, which a launcher may not hand to every rank alike) by a MIN all-reduce here, the availability of librccl
        # and the outcome of ncclCommInitRank inside init_native_comm_from_process_group -- never decided per rank.
        wish = os.environ.get("LK_NATIVE_RCCL", "1")
        want = torch.tensor([1 if (wish == "force" or (backend == "nccl" and wish != "0")) else 0], dtype=torch.int32,
                            device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(want, op=dist.ReduceOp.MIN)
        native = bool(int(want.item()))
        if native:
            progress.phase("native RCCL communicator: agreeing on librccl, unique id, ncclCommInitRank")
            if hook("LK_TEST_FAIL_COMM_RANK") == str(rank):               # test hook: this rank alone cannot enter the collective
                _die(progress, "LK_TEST_FAIL_COMM_RANK: simulated failure of lk_comm_init_rank on this rank only")
            try:
                native = ctx.init_native_comm_from_process_group(dist.group.WORLD)
            except Exception as exc:  # noqa: BLE001 - agreed on by all ranks inside; every rank leaves
                _die(progress, f"native RCCL communicator failed: {exc!r}")
            if native:
                reduce_path = "RCCL native (ncclAllReduce issued by liblightkrylov_hip on its own stream)"
            elif rank == 0:
                print("bench.py: librccl is not available on every rank; all ranks use torch.distributed", file=sys.stderr, flush=True)
        if not native:
            ctx.set_process_group(dist.group.WORLD)
            reduce_path = ("RCCL" if backend == "nccl" else backend) + " via torch.distributed callback"
        progress.phase(f"communicator up: {reduce_path}")
        if hook("LK_TEST_SCALE_PARTIALS_RANK") == str(rank):
            # test hook: THIS rank's contribution to every sweep reduction is scaled by 1 + 1e-6 before the sum (callback route only) --
            # a wrong reduction; the line's parity check must catch it and the run must end non-zero
            if native:
                _die(progress, "LK_TEST_SCALE_PARTIALS_RANK needs the callback route (LK_NATIVE_RCCL=0)")
            _install_faulty_reduction(ctx, torch, 1.0 + 1e-6)
    if args.grid_mult:
        ctx.set_tuning("grid_mult", args.grid_mult)
    for kv in args.tune:
        key, val = kv.split("=")
        ctx.set_tuning(key, int(val))

    dtype = np.float64 if args.dtype == "f64" else np.complex128
    if args.n is None:
        args.n = {"diag": 100_000_000, "dense": 65_536, "lap5": 4096 * 4096, "csr": 4096 * 4096}[args.operator]
    n, m = args.n, args.m
    s = 8 if args.dtype == "f64" else 16
    if args.operator != "diag" and args.dtype != "f64":
        raise SystemExit("bench.py: --operator dense / lap5 / csr run in real(dp)")
    N = int(round(n ** 0.5))
    if args.operator in ("lap5", "csr"):
        if N * N != n:
            raise SystemExit("bench.py: --operator lap5 / csr need --rows = N^2")
        j0, nj = lk.grid_partition(N, world, rank)          # whole grid lines per rank
        row0, n_local = j0 * N, nj * N
        row_starts = [lk.grid_partition(N, world, r)[0] * N for r in range(world)] + [n]
    else:
        row0, n_local = lk.row_partition(n, world, rank)
        row_starts = [lk.row_partition(n, world, r)[0] for r in range(world)] + [n]
    if args.shard_of:
        if world != 1 or args.operator != "diag":
            raise SystemExit("bench.py: --shard-of runs in one process on the diagonal operator")
        row0, n_local = lk.row_partition(n, args.shard_of, 0)       # rank 0's block of the P-rank job, alone on this GPU
    ctx.set_partition(row0, n)

    # ---- inputs resident in HBM before the timed region
    progress.phase(f"generating the inputs in HBM ({args.operator} operator, n_local = {n_local}, m = {m})")
    X = lk.krylov_basis_gpu(n_local, m + 1, dtype, ctx)
    keep = None
    if args.operator == "diag":
        if args.dtype == "f64":
            A = lk.diag_linop_gpu(n_local=n_local, row0=row0, d0=1.0, dstep=1.0 / n, ctx=ctx)
        else:
            g = (row0 + np.arange(n_local)) / n
            A = lk.diag_linop_gpu(((1.0 + g) * np.exp(1j * g)).astype(dtype), ctx)
        op_desc = "synthetic diagonal linop d_i=1+i/n"
        mv_bytes_model = (2 if args.dtype == "f64" else 3) * s * n_local
    elif args.operator == "dense":
        # A(i, j) = 2 u(seed = j, counter = i) - 1 generated in HBM column by column (the shared counter RNG: identical for
        # every partition); this rank holds rows [row0, row0 + n_local)
        keep = lk.krylov_basis_gpu(n_local, n, dtype, ctx)
        for j in range(n):
            keep[j].rand(False, seed=1000 + j)
        A = lk.dense_linop_gpu.from_device_panel(keep, n_global=n, row_starts=row_starts)
        op_desc = "synthetic dense linop A_ij = 2u(j, i)-1 (GEMV; row block per rank + all-gather of x)"
        mv_bytes_model = s * n_local * n + s * (n + n_local)
    elif args.operator == "lap5":
        A = lk.laplacian2d_linop_gpu(N, ctx, j0=j0, nj=nj) if world > 1 else lk.laplacian2d_linop_gpu(N, ctx)
        op_desc = f"5-point Laplacian stencil, {N} x {N} grid (grid lines per rank + one-line halo exchange)"
        mv_bytes_model = 2 * s * n_local
    else:
        A = lk.csr_linop_gpu(_laplacian_csr_rows(N, row0, n_local), ctx, n_global=n, row_starts=row_starts)
        op_desc = f"5-point Laplacian as a CSR matrix, {N} x {N} grid, {A.nnz} non-zeros on rank 0 (row block per rank + all-gather of x)"
        mv_bytes_model = (s + 4) * A.nnz + 8 * (n_local + 1) + s * (n + n_local)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")

    def one_factorisation() -> int:
        X[0].rand(True, seed=7)          # x0_i = 2u(i)-1, normalised (global norm via all-reduce)
        return lk.arnoldi(A, X, H)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    progress.phase(f"warm-up: {args.warmup} factorisation(s)")
    for _ in range(args.warmup):
        one_factorisation()
        progress.tick()
    fence()
    ctx.profile_reset()
    ctx.profile_enable(not args.no_profile)
    fence()
    progress.phase(f"timed region: {args.steps} factorisation(s)")
    t0 = time.perf_counter()
    info = 0
    for _ in range(args.steps):
        info = one_factorisation()
        progress.tick()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    progress.phase(f"timed region done ({elapsed:.2f} s on this rank); max over ranks")
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_sweeps, sweep_ms, sweep_bytes = ctx.profile_get("dgs_sweep*")
    n_dgs, dgs_ms, dgs_bytes = ctx.profile_get("dgs")
    n_mv, mv_ms, mv_bytes = ctx.profile_get("matvec")
    comm_tags = ("comm_allreduce", "comm_halo", "comm_allgather")
    comm_local = {t: ctx.profile_get(t) for t in comm_tags}
    # every rank's own numbers to rank 0: the line's roofline is the SLOWEST rank's, and the spread says whether a scaling gap is
    # rank skew (one slow rank; the others wait for it inside the all-reduce) or a uniform slowdown
    mine = [float(n_sweeps), sweep_ms, sweep_bytes, elapsed_local, float(n_mv), mv_ms, float(n_dgs), dgs_ms]
    for t in comm_tags:
        mine += [float(comm_local[t][0]), comm_local[t][1], comm_local[t][2]]
    per_rank_rows = [mine]
    if dist is not None and world > 1:
        tt = torch.tensor(mine, dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        every = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank_rows = [[float(v) for v in e.tolist()] for e in every]
    per_sweep = {}
    for i, what in ((1, "h1 = X^H y"), (2, "y' = y - X h1 (registers); h2 = X^H y'"), (3, "y'' = (y - X h1) - X h2, stored")):
        cnt, ms, by = ctx.profile_get(f"dgs_sweep{i}")
        if cnt:
            per_sweep[f"sweep{i}"] = {"computes": what, "launches": int(cnt), "avg_ms": ms / cnt,
                                      "algorithmic_bytes_per_launch": by / cnt,
                                      "GBps": by / ms / 1e6, "frac": by / ms / 1e6 / HBM_PEAK_GBS}

    if rank == 0 and args.dump_h:
        np.save(args.dump_h, H)
    if rank == 0:
        iters = args.steps * m
        achieved = (sweep_bytes / 1e9) / (sweep_ms / 1e3) if sweep_ms > 0 else 0.0
        # `traffic` comes from separate rocprofv3 --pmc passes (counters cannot be collected from inside this process); the
        # record names the SHA-256 of the kernel sources it was measured on.  A record made on other kernels is refused
        # loudly -- traffic = null and a line on stderr -- instead of being quoted as if it described this build.
        traffic, traffic_src = load_traffic_record(n_local, m, args.dtype) if args.operator == "diag" else (None, "no PMC record for this workload")
        # roofline of the SLOWEST rank (the one the others wait for); this rank's own figures sit beside it
        R = per_rank_rows
        rank_avg = [r[1] / max(r[0], 1.0) for r in R]                          # avg launch ms per rank
        rank_gbs = [(r[2] / 1e9) / (r[1] / 1e3) if r[1] > 0 else 0.0 for r in R]
        slow = int(np.argmax(rank_avg)) if R else 0
        achieved_rank0 = achieved
        if world > 1 and rank_gbs[slow] > 0:
            achieved = rank_gbs[slow]
        per_rank = {
            "avg_launch_ms": {"min": min(rank_avg), "max": max(rank_avg), "mean": float(np.mean(rank_avg)), "by_rank": rank_avg},
            "GBps": {"min": min(rank_gbs), "max": max(rank_gbs), "mean": float(np.mean(rank_gbs))},
            "launches_by_rank": [int(r[0]) for r in R],
            "timed_region_s_by_rank": [r[3] for r in R],
            "slowest_rank": slow, "frac_is": "the slowest rank's",
        }
        # collectives on the engine's stream (HIP events either side of each one): what N > 1 adds to a step
        comm = {"route": reduce_path, "ranks": world}
        comm_ms_rank0 = 0.0
        for ti, tag in enumerate(comm_tags):
            cnt0, ms0, by0 = comm_local[tag]
            comm_ms_rank0 += ms0
            col = 8 + 3 * ti
            ms_by_rank = [r[col + 1] for r in R]
            comm[tag[len("comm_"):]] = {
                "launches": int(cnt0), "avg_us": (1e3 * ms0 / cnt0) if cnt0 else None, "ms_total": ms0,
                "bytes_per_launch": (by0 / cnt0) if cnt0 else None,
                "frac_of_step_time": (ms0 / 1e3) / elapsed if elapsed > 0 else 0.0,
                "ms_total_by_rank": {"min": min(ms_by_rank), "max": max(ms_by_rank), "mean": float(np.mean(ms_by_rank))},
            }
        comm["frac_of_step_time"] = (comm_ms_rank0 / 1e3) / elapsed if elapsed > 0 else 0.0
        comm["measured"] = ("HIP events recorded on the engine's stream before and after every collective (native route: the time the RCCL kernel holds "
                            "the stream = launch + ring latency + the wait for the slowest peer; host routes: the whole round trip); rank 0's totals, "
                            "min / max / mean over ranks beside them") if world > 1 or dist is not None else "single rank: no collective is issued"
        # where the step time of rank 0 went: sweeps + operator + collectives + the rest (finish kernels, normalise, launch gaps, host)
        attribution = {"sweeps": (sweep_ms / 1e3) / elapsed, "matvec": (mv_ms / 1e3) / elapsed, "comm": comm["frac_of_step_time"]} if elapsed > 0 else {}
        if attribution:
            attribution["other (finish kernels, normalise, launch gaps, host)"] = max(0.0, 1.0 - sum(attribution.values()))
        # self-check of the line: where the repository holds a fixture of the reference's arithmetic for exactly this workload (the
        # metric configuration and configs[1]), every column of the H just computed is compared with it -- whatever the number of
        # ranks, so a scaling run whose reduction went wrong says so in its own line (the fixture is data, tests/golden/)
        parity = None
        fx = os.path.join(ROOT, "tests", "golden", f"arnoldi_diaglin_n{n}_m{m}_rdp.npz")
        if args.operator == "diag" and args.dtype == "f64" and os.path.exists(fx):
            try:
                z = np.load(fx)
                cerr = lambda Ha, Hb: float(max(np.abs(Ha[:, j] - Hb[:, j]).max() / np.abs(Hb[:, j]).max() for j in range(Hb.shape[1])))  # noqa: E731
                e_seq, e_comp = cerr(H, z["H_seq"]), cerr(H, z["H_comp"])
                parity = {"fixture": os.path.relpath(fx, ROOT), "max_normwise_column_error_vs_reference_arithmetic": e_seq,
                          "vs_twice_working_precision_dots": e_comp, "bound": 1e-12, "ok": bool(e_seq <= 1e-12)}
            except Exception as exc:  # noqa: BLE001
                parity = {"fixture": os.path.relpath(fx, ROOT), "ok": None, "error": repr(exc)}
        out = {
            "metric": "Arnoldi iterations/s (+ DGS sweep HBM GB/s, % of 8 TB/s roofline)",
            "value": iters / elapsed,
            "unit": "Arnoldi iterations/s",
            "n_gpus": world,
            "gpus_requested": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"arnoldi, {op_desc}, n={n} {'real' if s == 8 else 'complex'}(dp), "
                            f"m={m}, one step = one m-step factorisation",
                "operator": args.operator,
                "n_global": n, "n_local": n_local, "m": m, "parallelism": f"row-shard x{world} (RCCL all-reduce of <=129 scalars/sweep)",
                "info": int(info), "H_fro": float(np.linalg.norm(H)), "H_last_subdiag": float(abs(H[m, m - 1])),
                "parity": parity,
                "all_reduce": reduce_path,
                "nccl_algo": os.environ.get("NCCL_ALGO") if world > 1 or dist is not None else None,
            },
            "roofline": {
                "bound": "hbm", "kernel": "the three DGS sweeps: lk::panel_dot_cw (DOT, one column at a time) | lk::panel_sweep (UPDATE+DOT, y' kept in registers) | lk::panel_sweep (UPDATE with two coefficient sets)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_measured_in_this_run": False,      # a stored record of separate --pmc passes (profiles/pmc_traffic.json)
                "traffic_source": traffic_src,
                "traffic_over_algorithmic": (traffic / (sweep_bytes / max(n_sweeps, 1))) if traffic and n_sweeps else None,
                "bytes_priced": "ALGORITHMIC three-sweep schedule, s*n_local*(k+1 | k+2 | k+2) = s*n*(3k+5) per DGS (SURVEY 8d); "
                                "the shipped schedule moves 3k+4 columns (y' is never written), so HBM-level bandwidth is "
                                "(3k+4)/(3k+5) of `achieved`",
                "launches": int(n_sweeps), "avg_launch_ms": rank_avg[slow] if world > 1 else sweep_ms / max(n_sweeps, 1),
                "rank0": {"achieved": achieved_rank0, "avg_launch_ms": sweep_ms / max(n_sweeps, 1)},
                "per_rank": per_rank,
                "algorithmic_bytes_per_launch": sweep_bytes / max(n_sweeps, 1),
                "per_sweep": per_sweep,
                "dgs_call_GBps": (dgs_bytes / 1e9) / (dgs_ms / 1e3) if dgs_ms > 0 else 0.0,
                "dgs_frac_of_step_time": (dgs_ms / 1e3) / elapsed if elapsed > 0 else 0.0,
                "matvec": {"launches": int(n_mv), "ms_total": mv_ms, "avg_ms": mv_ms / max(n_mv, 1),
                           "algorithmic_bytes_per_launch": mv_bytes_model,
                           "GBps": (mv_bytes_model * n_mv / mv_ms / 1e6) if mv_ms > 0 else None,
                           "frac_of_hbm_peak": (mv_bytes_model * n_mv / mv_ms / 1e6 / HBM_PEAK_GBS) if mv_ms > 0 else None,
                           "frac_of_step_time": (mv_ms / 1e3) / elapsed if elapsed > 0 else 0.0,
                           "measured": ("HIP events around the operator (its own dispatch for the one-kernel diagonal operators; stream "
                                        "markers around kernels + exchange otherwise) inside the asynchronous batch") if n_mv else "not measured"},
            },
        }
        out["comm"] = comm
        out["step_time_attribution_rank0"] = attribution
        pred = predicted_iters_per_s(n, m, args.shard_of if args.shard_of else world, args.operator, args.dtype)
        out["predicted_it_s"] = pred["predicted_it_s"] if pred else None
        out["predicted"] = pred
        if pred and pred.get("predicted_it_s"):
            out["predicted"]["value_over_predicted"] = out["value"] / pred["predicted_it_s"]
        if args.shard_of:
            out["shard_emulation"] = {"of_ranks": args.shard_of, "note": "ONE row block of a P-rank job alone on this GPU (no communicator): "
                                      "per-shard kernel measurements only -- `value` is NOT the metric, it is what the P-rank job would make with free collectives; "
                                      "`predicted` is the model for that P-rank job"}
            out["config"]["parity"] = None
        if mv_ms > sweep_ms and mv_ms > 0:
            # the operator, not the orthogonalisation, is the dominant kernel of this workload (dense GEMV): its roofline leads
            r = out["roofline"]
            r["dominant"] = "matvec"
            r["dgs_sweeps"] = {"achieved": r["achieved"], "frac": r["frac"], "launches": r["launches"], "avg_launch_ms": r["avg_launch_ms"],
                               "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"]}
            r["kernel"] = {"dense": "lk::k_gemv_n (y = A_rows x: 16 B per lane along rows, 8 columns in flight, split-K in chunks of 256 columns) + lk::k_gemv_n_finish"}.get(args.operator, "operator")
            r["achieved"] = r["matvec"]["GBps"]
            r["frac"] = r["matvec"]["frac_of_hbm_peak"]
            r["launches"], r["avg_launch_ms"] = int(n_mv), mv_ms / max(n_mv, 1)
            r["algorithmic_bytes_per_launch"] = mv_bytes_model
            r["bytes_priced"] = "ALGORITHMIC bytes of one operator application: s*n_local*n (the matrix) + s*(n + n_local) (x read, y written)"
            r["traffic"], r["traffic_source"] = None, "no PMC record for this workload"
        if world == 1 and not args.no_cpu_baseline:
            progress.phase("cpu_baseline: the oracle on this host's cores (bounded samples)")
        if world == 1 and not args.no_cpu_baseline and args.operator != "diag":
            try:
                out["cpu_baseline"] = cpu_baseline_operator(args.operator, n, m)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Arnoldi iterations/s", "cores": 1, "kind": "port", "sample": f"failed: {exc!r}"}
        elif world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, m, args.cpu_n, args.cpu_m, progress)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Arnoldi iterations/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {exc!r}"}
        # (a context leg like cpu_baseline: --no-cpu-baseline drops it too -- the PMC / kernel-trace passes of tools/run_profiles_final.sh must see the
        #  metric workload's launches only, the small-n sweeps would land in the same kernel instantiations' averages)
        if world == 1 and args.operator == "diag" and not args.shard_of and not args.no_cpu_baseline:
            try:
                out["launch_bound_regime"] = launch_bound_regime(ctx)
            except Exception as exc:  # noqa: BLE001 - context only: never costs the metric line
                out["launch_bound_regime"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
        parity_failed = bool(out["config"].get("parity")) and out["config"]["parity"].get("ok") is False
    # a line whose own parity check failed (a reduction that went wrong in a sharded run) is printed -- and the run ends NON-ZERO on
    # every rank: a throughput number for a wrong factorisation is not a result
    if dist is not None and world > 1:
        flag = torch.tensor([1 if (rank == 0 and parity_failed) else 0], dtype=torch.int32, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        parity_failed = bool(int(flag.item()))
    # orderly teardown: device objects, then the library's own communicator (lk_finalize), then torch's group
    progress.phase("teardown")
    del X, A, keep
    if dist is not None:
        dist.barrier()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    progress.done()
    if parity_failed:
        print(f"bench.py[rank {rank}/{world}]: config.parity.ok is false -- the Hessenberg matrix of this run is not the reference's; exit 4", file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    main()
