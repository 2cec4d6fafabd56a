"""Block Arnoldi (arnoldi.fypp:20-73, blksize = p) as ONE asynchronous engine call (lk_arnoldi_block) against one host round trip per step
("async_arnoldi" = 0: lk_dgs_block + lk_qr per step), interleaved in one process; diagonal operator.
  python tools/bench_block_arnoldi.py [f64|c128]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
dtype = np.complex128 if len(sys.argv) > 1 and sys.argv[1] == "c128" else np.float64
ctx = lk.Context(device=0)
for n, p, kdim in ((175_000, 2, 32), (175_000, 4, 16), (1_000_000, 2, 32), (1_000_000, 4, 16), (1_000_000, 8, 16), (10_000_000, 4, 16)):
    nn = n if dtype is np.float64 else n // 2
    X = lk.krylov_basis_gpu(nn, (kdim + 1) * p, dtype, ctx)
    A = lk.diag_linop_gpu(n_local=nn, row0=0, d0=1.0, dstep=1.0 / nn, ctx=ctx) if dtype is np.float64 else lk.diag_linop_gpu((1.0 + np.arange(nn) / nn).astype(dtype), ctx)
    H = np.zeros(((kdim + 1) * p, kdim * p), dtype=dtype, order="F")
    best = {0: 1e9, 1: 1e9}
    for rep in range(5):
        for mode in (0, 1):
            ctx.set_tuning("async_arnoldi", mode)
            for j in range(p):
                X[j].rand(True, seed=7 + j)
            R = np.zeros((p, p), dtype=dtype, order="F")
            lk.qr(X[:p], R)
            ctx.sync()
            t0 = time.perf_counter()
            assert lk.arnoldi(A, X, H, blksize=p) == 0
            dt = time.perf_counter() - t0
            if rep: best[mode] = min(best[mode], dt)
    print(json.dumps({"dtype": np.dtype(dtype).name, "n": nn, "blksize": p, "kdim": kdim, "block_steps_per_s": {"one_round_trip_per_step": round(kdim / best[0], 1),
                      "asynchronous": round(kdim / best[1], 1)}, "ms_per_block_step": [round(best[0] / kdim * 1e3, 3), round(best[1] / kdim * 1e3, 3)]}), flush=True)
    del X
