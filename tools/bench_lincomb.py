"""krylov_schur-shaped basis update X Z on the GPU (lk_lincomb -> lk::panel_gemm): ms per call, FP64 TFLOP/s and
GB/s on the algorithmic k+q columns.  The complex kind is FMA-bound (8kq flop per 16(k+q) bytes), the real kind
about balanced; FP64 vector/matrix peak of MI355X is 78.6 TFLOP/s."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
ctx = lk.Context(device=0)
for kv in sys.argv[1:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
shapes = ((np.complex128, 1_000_000, 128, 64), (np.complex128, 1_000_000, 128, 16), (np.complex128, 1_000_000, 64, 32),
          (np.float64, 10_000_000, 128, 64), (np.float64, 10_000_000, 64, 32), (np.float64, 10_000_000, 128, 1))
if os.environ.get("LK_LINCOMB_SCAN"):      # narrow products: q = 1..32 for the crossover table (set gemm_mfma_min=1 / 100 on the command line)
    shapes = tuple((dt, nn, k, q) for dt, nn in ((np.float64, 10_000_000), (np.complex128, 5_000_000)) for k in (32, 64, 128)
                   for q in (1, 2, 3, 4, 5, 8, 16, 32))
if os.environ.get("LK_LINCOMB_SHAPE"):     # one shape: "f64,10000000,64,32"
    dt, nn, kk, qq = os.environ["LK_LINCOMB_SHAPE"].split(",")
    shapes = ((np.float64 if dt == "f64" else np.complex128, int(nn), int(kk), int(qq)),)
for dtype, n, k, q in shapes:
    X = lk.krylov_basis_gpu(n, k, dtype, ctx)
    for j in range(k):
        X[j].rand(True, seed=j)
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((k, q)) + (1j * rng.standard_normal((k, q)) if np.dtype(dtype).kind == "c" else 0)
    Z = np.asfortranarray(Z.astype(dtype))
    lk.linear_combination(X, Z)
    ctx.profile_reset(); ctx.profile_enable(True)
    ctx.sync(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        Y = lk.linear_combination(X, Z)
    ctx.sync(); dt = (time.perf_counter() - t0) / reps
    cnt, ms, by = ctx.profile_get("lincomb"); ctx.profile_enable(False)
    s = np.dtype(dtype).itemsize
    flop = (8.0 if s == 16 else 2.0) * n * k * q
    print(json.dumps({"dtype": str(np.dtype(dtype)), "n": n, "k": k, "q": q, "launches_per_call": cnt / reps,
                      "kernel_ms_per_call": ms / reps, "wall_ms_per_call": dt * 1e3,
                      "TFLOPs_fp64": flop / (ms / reps) / 1e9, "frac_of_78.6TF": flop / (ms / reps) / 1e9 / 78.6,
                      "GBps_algorithmic(k+q cols)": s * n * (k + q) / (ms / reps) / 1e6}), flush=True)
    del X, Y
