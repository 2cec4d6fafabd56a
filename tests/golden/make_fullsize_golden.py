"""Generator of the full-size golden fixtures (run on the GPU box, whose host has the cores and the 104 GB of RAM):
the multi-threaded oracle in SEQUENTIAL mode (= the reference's arithmetic, bit-identical to the 1-thread
restatement) and in COMPENSATED mode (twice-working-precision dots) on the diag-linspace Arnoldi workload of
BASELINE configs[1] / configs[4], with the engine's own run beside them for the report.

  python tests/golden/make_fullsize_golden.py --rows 100000000 --kdim 128 --out gpurun_out/r02/fullsize_n1e8_m128.npz

Writes H_seq, H_comp (what tests/golden/arnoldi_diaglin_*.npz keep), H_gpu (this run's engine output, for the
report only) and a JSON `meta` string.  Test infrastructure: it imports the oracle, so it lives under tests/."""
import argparse, json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--kdim", type=int, default=128)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--out", required=True)
ap.add_argument("--no-gpu", action="store_true")
args = ap.parse_args()
n, m = args.rows, args.kdim


def colerr(A, B):
    return max(np.abs(A[:, j] - B[:, j]).max() / np.abs(B[:, j]).max() for j in range(B.shape[1]))


def ritz(H):
    w = np.linalg.eigvals(H[:m, :m])
    return w[np.argsort(w.real, kind="stable")]


def ritzerr(Ha, Hb):
    a, b = ritz(Ha), ritz(Hb)
    return float(np.max(np.abs(a - b) / np.abs(b)))


meta = {"n": n, "m": m, "seed": args.seed, "operator": "d_i = fma(1/n, i, 1.0)", "x0": "2u(i)-1, seed 7, normalised",
        "host": os.uname().nodename}
try:
    meta["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:  # noqa: BLE001
    meta["commit"] = os.environ.get("LK_COMMIT", "unknown")

H_gpu = None
if not args.no_gpu:
    import lightkrylov_amd as lk
    ctx = lk.Context(device=0)
    X = lk.krylov_basis_gpu(n, m + 1, np.float64, ctx)
    A = lk.diag_linop_gpu(n_local=n, row0=0, d0=1.0, dstep=1.0 / n, ctx=ctx)
    H_gpu = np.zeros((m + 1, m), order="F")
    X[0].rand(True, seed=args.seed)
    t0 = time.perf_counter()
    info = lk.arnoldi(A, X, H_gpu)
    ctx.sync()
    meta["gpu_seconds"] = time.perf_counter() - t0
    meta["gpu_info"] = int(info)
    G = lk.Gram(X[:m + 1]) if hasattr(lk, "Gram") else None
    if G is not None:
        meta["gpu_orth"] = float(np.abs(G - np.eye(m + 1)).max())
    del X, A
    ctx.close()
    print("gpu done", meta.get("gpu_seconds"), meta.get("gpu_orth"), flush=True)

from oracle import oracle as ora
T = args.threads or ora.max_threads()
T = ora.set_threads(T)
meta["oracle_threads"] = T
Xo = np.zeros((n, m + 1), order="F")
out = {}
for name, mode in (("seq", ora.SEQUENTIAL), ("comp", ora.COMPENSATED)):
    ora.fill_counter(Xo[:, 0], args.seed)
    # x0 normalised with the mode's own norm (the reference: x0%scal(1/x0%norm()))
    nrm = float(np.sqrt(abs(ora.dot_mode(Xo[:, 0], Xo[:, 0], mode))))
    ora.scal(Xo[:, 0], 1.0 / nrm)
    H = np.zeros((m + 1, m), order="F")
    t0 = time.perf_counter()
    info = ora.arnoldi(ora.DiagLinOp(1.0, 1.0 / n), Xo, H, fast=True, mode=mode)
    meta[f"oracle_{name}_seconds"] = time.perf_counter() - t0
    meta[f"oracle_{name}_info"] = int(info)
    out["H_" + name] = H
    print(name, "done", meta[f"oracle_{name}_seconds"], flush=True)

meta["seq_vs_comp_H"] = colerr(out["H_seq"], out["H_comp"])
meta["seq_vs_comp_ritz"] = ritzerr(out["H_seq"], out["H_comp"])
if H_gpu is not None:
    out["H_gpu"] = H_gpu
    for name in ("seq", "comp"):
        meta[f"gpu_vs_{name}_H"] = colerr(H_gpu, out["H_" + name])
        meta[f"gpu_vs_{name}_ritz"] = ritzerr(H_gpu, out["H_" + name])
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
np.savez_compressed(args.out, meta=json.dumps(meta), **out)
print(json.dumps(meta))
