"""One size of the double Gram-Schmidt step, a few repetitions: the target of `rocprofv3 --pmc ... -- python3 tools/dgs_once.py n k [reps]`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
n, k = int(float(sys.argv[1])), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = lk.Context(device=0)
for kv in sys.argv[4:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val))
B = lk.krylov_basis_gpu(n, k + 1, np.float64, ctx)
for j in range(k + 1):
    B[j].rand(True, seed=100 + j)
for _ in range(reps):
    lk.double_gram_schmidt_step(B[k], B[:k], False)
ctx.sync()
print("done", n, k, reps, ctx.resident_stats())
