#!/usr/bin/env python3
"""bench.py -- Arnoldi-iteration throughput + DGS sweep bandwidth on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is ONE Arnoldi factorisation of the workload (m Arnoldi iterations: operator kernel,
three fused DGS panel sweeps, normalise), run through the C ABI (lk_arnoldi).  Workload at every N:
BASELINE.json's metric configuration -- synthetic diagonal operator d_i = 1 + i/n, n = 10^8
real(dp) rows, m = 128, x0 from the shared counter RNG -- row-sharded over the N ranks (strong
scaling: total work fixed), the only cross-rank traffic being the RCCL all-reduce of the <= 129
reduction scalars after each sweep.  Inputs are generated in HBM before the timed region.

Prints ONE JSON line on rank 0; `value` = Arnoldi iterations per second, whole job.
  roofline   : the panel sweep kernel (lk::panel_sweep, three instantiations) -- algorithmic bytes s*n_local*(k+1|k+2)
               per launch (s*n*(3k+5) per DGS) / HIP-event duration on the kernel's
               stream, averaged over every sweep launch of the timed region; peak = 8 TB/s HBM3E.
  cpu_baseline: the reference-schedule CPU oracle (oracle/, 1 thread, kind "port") timed on this
               host on a bounded sample and scaled to the metric's unit (see `sample`).
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


def cpu_baseline(n_full: int, m_full: int, budget_n: int, budget_m: int) -> dict:
    """Time the oracle's Arnoldi (reference schedule: per-primitive BLAS-1 calls, sequential dots,
    scal-then-axpy axpby, fresh projection vector per pass; single thread like the reference) on a
    bounded sample and scale to the full workload with the reference's own traffic model
    (14k+20)*s*n bytes per step."""
    from oracle import oracle as ora
    n, m = budget_n, budget_m
    d = 1.0 + np.arange(n) / n
    X = np.zeros((n, m + 1), order="F")
    ora.fill_counter(X[:, 0], 7)
    ora.scal(X[:, 0], 1.0 / ora.norm(X[:, 0]))
    H = np.zeros((m + 1, m), order="F")
    t0 = time.perf_counter()
    info = ora.arnoldi(ora.DiagOp(d), X, H)
    dt = time.perf_counter() - t0
    model = lambda nn, mm: sum(8.0 * nn * (14 * k + 20 + 3) for k in range(1, mm + 1))  # noqa: E731  (+3: diag matvec)
    bw = model(n, m) / dt
    t_full = model(n_full, m_full) / bw
    return {
        "value": m_full / t_full, "unit": "Arnoldi iterations/s", "cores": 1, "kind": "port",
        "measured_sample_iters_per_s": m / dt, "measured_sample_seconds": dt, "info": int(info),
        "effective_GBps_on_reference_schedule": bw / 1e9,
        "host_cpus": os.cpu_count(),
        "sample": (f"oracle (C restatement of the reference schedule, 1 thread) arnoldi, diagonal operator, "
                   f"n={n}, m={m}, real(dp): {dt:.2f} s measured; scaled to n={n_full}, m={m_full} by the "
                   f"reference schedule's bytes, sum_k 8n(14k+23)"),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=100_000_000, help="global rows (metric config: 1e8)")
    ap.add_argument("--kdim", dest="m", type=int, default=128, help="Krylov dimension (metric config: 128)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "c128"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=8_000_000)
    ap.add_argument("--cpu-m", type=int, default=24)
    ap.add_argument("--grid-mult", type=int, default=0)
    args = ap.parse_args()

    import torch
    import lightkrylov_amd as lk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_distributed.py): several ranks on ONE device with the gloo backend, which
    # all-reduces CUDA tensors through the host -- exercises the multi-process sharded path where RCCL cannot
    # (RCCL refuses two ranks on one GPU).  Never set by the driver.
    backend = os.environ.get("LK_DIST_BACKEND", "nccl")
    if "LK_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["LK_FORCE_DEVICE"])
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run: RCCL path even for one rank
        import torch.distributed as dist  # noqa: PLW0621
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    ctx = lk.Context(device=local_rank)
    if dist is not None:
        ctx.set_process_group(dist.group.WORLD)
    if args.grid_mult:
        ctx.set_tuning("grid_mult", args.grid_mult)

    dtype = np.float64 if args.dtype == "f64" else np.complex128
    n, m = args.n, args.m
    row0, n_local = lk.row_partition(n, world, rank)
    ctx.set_partition(row0, n)
    s = 8 if args.dtype == "f64" else 16

    # ---- inputs resident in HBM before the timed region
    X = lk.krylov_basis_gpu(n_local, m + 1, dtype, ctx)
    if args.dtype == "f64":
        A = lk.diag_linop_gpu(n_local=n_local, row0=row0, d0=1.0, dstep=1.0 / n, ctx=ctx)
    else:
        g = (row0 + np.arange(n_local)) / n
        A = lk.diag_linop_gpu(((1.0 + g) * np.exp(1j * g)).astype(dtype), ctx)
    H = np.zeros((m + 1, m), dtype=dtype, order="F")

    def one_factorisation() -> int:
        X[0].rand(True, seed=7)          # x0_i = 2u(i)-1, normalised (global norm via all-reduce)
        return lk.arnoldi(A, X, H)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        one_factorisation()
    fence()
    ctx.profile_reset()
    ctx.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    info = 0
    for _ in range(args.steps):
        info = one_factorisation()
    fence()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_sweeps, sweep_ms, sweep_bytes = ctx.profile_get("dgs_sweep*")
    n_dgs, dgs_ms, dgs_bytes = ctx.profile_get("dgs")
    n_mv, mv_ms, _ = ctx.profile_get("matvec")

    if rank == 0:
        iters = args.steps * m
        achieved = (sweep_bytes / 1e9) / (sweep_ms / 1e3) if sweep_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("n_local") == n_local and rec.get("m") == m and rec.get("dtype") == args.dtype:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "Arnoldi iterations/s (+ DGS sweep HBM GB/s, % of 8 TB/s roofline)",
            "value": iters / elapsed,
            "unit": "Arnoldi iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"arnoldi, synthetic diagonal linop d_i=1+i/n, n={n} {'real' if s == 8 else 'complex'}(dp), "
                            f"m={m}, one step = one m-step factorisation",
                "n_global": n, "n_local": n_local, "m": m, "parallelism": f"row-shard x{world} (RCCL all-reduce of <=129 scalars/sweep)",
                "info": int(info), "H_fro": float(np.linalg.norm(H)), "H_last_subdiag": float(abs(H[m, m - 1])),
                "all_reduce": (("RCCL" if backend == "nccl" else backend) + " via torch.distributed") if dist is not None else "none (single rank)",
            },
            "roofline": {
                "bound": "hbm", "kernel": "lk::panel_sweep, the three DGS sweeps (DOT | UPDATE+DOT, y' kept in registers | UPDATE with two coefficient sets)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launches": int(n_sweeps), "avg_launch_ms": sweep_ms / max(n_sweeps, 1),
                "algorithmic_bytes_per_launch": sweep_bytes / max(n_sweeps, 1),
                "dgs_call_GBps": (dgs_bytes / 1e9) / (dgs_ms / 1e3) if dgs_ms > 0 else 0.0,
                "dgs_frac_of_step_time": (dgs_ms / 1e3) / elapsed if elapsed > 0 else 0.0,
                "matvec_ms_total": mv_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, m, args.cpu_n, args.cpu_m)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Arnoldi iterations/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {exc!r}"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
