"""HBM bandwidth of the abstract_vector primitives (the six type-bound procedures + norm / copy), one launch each, on
algorithmic bytes (scal 2 s n | axpby 3 s n | dot 2 s n | norm s n | copy 2 s n | zero, rand s n)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
for dtype, s in ((np.float64, 8), (np.complex128, 16)):
    nn = n if s == 8 else n // 2
    c = lk.Context(device=0)
    B = lk.krylov_basis_gpu(nn, 3, dtype, c)
    x, y, z = B[0], B[1], B[2]
    x.rand(False, seed=1); y.rand(False, seed=2)
    ops = {
        "scal": (lambda: x.scal(1.0000001), 2), "axpby": (lambda: y.axpby(0.5, x, 0.999), 3),
        "axpby_beta0 (copy semantics)": (lambda: z.axpby(1.0, x, 0.0), 2),
        "dot": (lambda: x.dot(y), 2), "norm": (lambda: x.norm(), 1), "copy": (lambda: lk.copy(z, x), 2),
        "zero": (lambda: z.zero(), 1), "rand": (lambda: z.rand(False, seed=3), 1),
    }
    A = lk.diag_linop_gpu(np.linspace(1, 2, nn).astype(dtype), c)
    ops["matvec diag (3 streams)"] = (lambda: A.matvec(x, z), 3)
    if s == 8:
        A2 = lk.diag_linop_gpu(n_local=nn, row0=0, d0=1.0, dstep=1.0 / nn, ctx=c)
        ops["matvec diag_linspace (2 streams)"] = (lambda: A2.matvec(x, z), 2)
    for mult in ([int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2]):
      c.set_tuning("blas1_grid_mult", mult)
      out = {"kind": "f64" if s == 8 else "c128", "n": nn, "blocks_per_CU": mult}
      for name, (fn, cols) in ops.items():
        fn(); c.sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        c.sync(); dt = (time.perf_counter() - t0) / reps
        out[name] = {"ms": round(dt * 1e3, 4), "GBps": round(cols * s * nn / dt / 1e9), "frac_of_8TBps": round(cols * s * nn / dt / 8e12, 3)}
      print(json.dumps(out), flush=True)
    del x, y, z, B, A, ops; c.close()
