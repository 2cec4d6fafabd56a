import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
n = 100_000_000
ctx = lk.Context(device=0)
kmax = 16
B = lk.krylov_basis_gpu(n, kmax + 1, np.float64, ctx)
for j in range(kmax + 1):
    B[j].rand(True, seed=100 + j)
for k in (1, 2, 3, 4, 6, 8, 12, 16):
    lk.double_gram_schmidt_step(B[kmax], B[:k], False)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(5):
        lk.double_gram_schmidt_step(B[kmax], B[:k], False)
    out = {}
    for tag in ("dgs_sweep1", "dgs_sweep2", "dgs_sweep3", "dgs_sweep*", "dgs"):
        c, ms, by = ctx.profile_get(tag)
        out[tag] = round(by / ms / 1e6)
    ctx.profile_enable(False)
    print(json.dumps({"k": k, "GBps": out}), flush=True)
