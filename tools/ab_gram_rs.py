"""A/B of the row-split LDS-DMA Gram kernel (panel_gram_rs, "gram_rs") against the kernels it would replace: parity vs numpy at ragged sizes, then ms per call at n = 10^7.
  python tools/ab_gram_rs.py [rows] [check|time|both]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

n_big = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
ctx = lk.Context(device=0)

DT = np.complex128 if os.environ.get("COMPLEX") else np.float64          # COMPLEX=1: the complex kind (panel_gram_rs3m at 33..80 columns)

def gram(n, k, rs, seed=0):
    ctx.set_tuning("gram_rs", rs)
    B = lk.krylov_basis_gpu(n, k, DT, ctx)
    for j in range(k):
        B[j].rand(True, seed=seed + j)
    G = lk.Gram(B)
    X = B.download(0, k) if n <= 200_000 else None
    del B
    return G, X

if mode in ("check", "both"):
    worst = 0.0
    for n in (2, 31, 32, 33, 95, 1000, 12345, 100003, 8192 * 32 + 7):
        for k in (33, 40, 48, 49, 64, 65, 80, 81, 90, 96, 97, 112, 113, 120, 128):
            G1, X = gram(n, k, 1)
            if X is not None:
                U = np.triu(X.conj().T @ X)                # the reference mirrors the upper triangle WITHOUT conjugating (AbstractVectors.fypp:645-657)
                ref = U + np.triu(U, 1).T
                err = np.abs(G1 - ref).max() / max(1.0, np.abs(ref).max())
            else:
                G0, _ = gram(n, k, 0)
                err = np.abs(G1 - G0).max() / max(1.0, np.abs(G0).max())
            worst = max(worst, err)
            if err > 1e-12:
                print("MISMATCH", n, k, err, flush=True)
    print(json.dumps({"check": "gram_rs vs numpy / staged kernels", "worst_rel_err": worst}), flush=True)

if mode in ("time", "both"):
    cfgs = [int(x) for x in os.environ.get("GRS", "0,1,2,4").split(",")]
    for k in (40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128):
        row = {"n": n_big, "k": k}
        B = lk.krylov_basis_gpu(n_big // (2 if DT == np.complex128 else 1), k, DT, ctx)
        for j in range(k):
            B[j].rand(True, seed=10 + j)
        for rep in range(2):                     # (two rounds over the configurations: the second one is reported, the first warms clocks and caches alike)
            for rs in cfgs:
                ctx.set_tuning("gram_rs", rs)
                lk.Gram(B); ctx.sync()
                ctx.profile_reset(); ctx.profile_enable(True)
                for _ in range(10):
                    lk.Gram(B)
                ctx.sync()
                cnt, ms, by = ctx.profile_get("xhy_mfma"); ctx.profile_enable(False)
                row["ms_%d" % rs] = round(ms / cnt, 3)
        del B
        KP = (k + 15) // 16
        flop = KP * (KP + 1) / 2 * 512 * n_big * (1.5 if DT == np.complex128 else 1.0)      # (complex: three real products on n / 2 elements)
        best = min(v for kk, v in row.items() if kk.startswith("ms_") and kk != "ms_0")
        row["TFLOPs_best"] = round(flop / best / 1e9, 1); row["TBps_best"] = round(n_big * 8 * k / best / 1e9, 2)
        print(json.dumps(row), flush=True)
