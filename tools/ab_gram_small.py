import json, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/lightkrylov_amd") else os.environ["GRAFT_REPO_ROOT"])
import lightkrylov_amd as lk
DT = np.complex128 if os.environ.get("COMPLEX") else np.float64
ctx = lk.Context(device=0)
worst = 0.0
for n in (1, 2, 31, 32, 33, 95, 1000, 12345, 100003):
    for k in (5, 8, 15, 16, 17, 24, 31, 32):
        B = lk.krylov_basis_gpu(n // (2 if DT == np.complex128 and n > 1000000 else 1), k, DT, ctx)
        for j in range(k):
            B[j].rand(True, seed=j)
        X = B.download(0, k)
        for sm in (1, 0):
            ctx.set_tuning("gram_rs", sm)
            G = lk.Gram(B)
            U = np.triu(X.conj().T @ X); ref = U + np.triu(U, 1).T
            e = np.abs(G - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, e)
            if e > 1e-12: print("MISMATCH", n, k, sm, e)
        del B
print(json.dumps({"check": "gram k<=32", "worst": worst}))
n = 10_000_000
for k in (8, 16, 24, 32):
    B = lk.krylov_basis_gpu(n // (2 if DT == np.complex128 and n > 1000000 else 1), k, DT, ctx)
    for j in range(k):
        B[j].rand(True, seed=10 + j)
    row = {"n": n, "k": k}
    for rep in range(2):
        for sm in (0, 1):
            ctx.set_tuning("gram_rs", sm)
            for _ in range(5): lk.Gram(B)
            ctx.sync(); ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(10): lk.Gram(B)
            ctx.sync()
            cnt, ms, by = ctx.profile_get("xhy_mfma"); ctx.profile_enable(False)
            row["ms_%d" % sm] = round(ms / cnt, 3)
    row["TBps_1"] = round(n * 8 * k / row["ms_1"] / 1e9, 2); row["TBps_0"] = round(n * 8 * k / row["ms_0"] / 1e9, 2)
    print(json.dumps(row), flush=True)
    del B
