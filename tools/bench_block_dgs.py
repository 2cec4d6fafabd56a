"""Block double Gram-Schmidt (DGS_basis_against_basis, gram_schmidt.fypp:59-105) on the GPU: time per call and, under
`rocprofv3 --pmc FETCH_SIZE`, the passes over X it makes.  n = 10^7 real(dp), k = 64 and 128 basis columns, p = 4."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
ctx = lk.Context(device=0)
n = 10_000_000
p = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2:
    ctx.set_tuning("block_fused", int(sys.argv[2]))       # 1: dots + fused update/dot + two-coefficient update (3 passes); 0: 4 passes
for k in (64, 128):
    B = lk.krylov_basis_gpu(n, k + p, np.float64, ctx)
    for j in range(k + p):
        B[j].rand(True, seed=100 + j)
    beta = np.zeros((k, p), order="F")
    lk.double_gram_schmidt_step(B[k:k + p], B[:k], False, beta)          # warm-up (also makes Y orthogonal; timing is data independent)
    ctx.sync(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        lk.double_gram_schmidt_step(B[k:k + p], B[:k], False, beta)
    ctx.sync(); dt = (time.perf_counter() - t0) / reps
    xbytes = 8.0 * n * k
    print(json.dumps({"n": n, "k": k, "p": p, "block_fused": int(sys.argv[2]) if len(sys.argv) > 2 else 1, "ms_per_block_dgs": dt * 1e3, "X_GB": xbytes / 1e9,
                      "passes_over_X_if_at_6.4TBps": dt * 6.4e12 / xbytes}), flush=True)
    del B
