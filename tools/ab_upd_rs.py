"""A/B of the row-owner fused pass of the block Gram-Schmidt (panel_xhy_upd_rs, "upd_rs") against panel_xhy_upd_mfma: DGS_basis_against_basis (gram_schmidt.fypp:59-105), real kind,
17..32 right-hand sides: parity (coefficients and Y against numpy and against the other kernel) at ragged sizes, then ms per block step and per fused pass at n = 10^7.
  python tools/ab_upd_rs.py [rows] [check|time|both]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

n_big = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
ctx = lk.Context(device=0)

def step(n, k, p, rs, seed=3):
    ctx.set_tuning("upd_rs", rs)
    B = lk.krylov_basis_gpu(n, k + p, np.float64, ctx)
    for j in range(k + p):
        B[j].rand(True, seed=seed + j)
    R = np.zeros((k, k), order="F")
    assert lk.qr(B[:k], R) == 0
    X = B.download(0, k); Y = B.download(k, p)
    beta = np.zeros((k, p), order="F")
    assert lk.double_gram_schmidt_step(B[k:k + p], B[:k], False, beta) == 0
    Yo = B.download(k, p)
    del B
    return X, Y, beta, Yo

if mode in ("check", "both"):
    worst = 0.0
    for n in (40, 64, 95, 161, 1000, 12345, 100003, 8192 * 32 + 7):
        for k, p in ((128, 32), (128, 17), (120, 24), (100, 32), (64, 32), (40, 20), (16, 32), (7, 18)):
            if k + p > n:
                continue
            X, Y, b1, Y1 = step(n, k, p, 1)
            _, _, b0, Y0 = step(n, k, p, 0)
            h1 = X.T @ Y; Ya = Y - X @ h1; h2 = X.T @ Ya; Yb = Ya - X @ h2
            sc = max(1.0, np.abs(h1).max())
            e = max(np.abs(b1 - (h1 + h2)).max() / sc, np.abs(Y1 - Yb).max(), np.abs(b1 - b0).max() / sc, np.abs(Y1 - Y0).max())
            worst = max(worst, e)
            if e > 1e-12:
                print("MISMATCH", n, k, p, e, flush=True)
    print(json.dumps({"check": "upd_rs vs numpy and vs panel_xhy_upd_mfma", "worst_err": worst}), flush=True)

if mode in ("time", "both"):
    cfgs = [int(x) for x in os.environ.get("URS", "0,1").split(",")]
    for k, p in ((128, 32), (128, 24), (96, 32), (64, 32), (32, 32))[:int(os.environ.get("NSHAPES", "5"))]:
        row = {"n": n_big, "k": k, "p": p}
        B = lk.krylov_basis_gpu(n_big, k + p, np.float64, ctx)
        for j in range(k + p):
            B[j].rand(True, seed=100 + j)
        beta = np.zeros((k, p), order="F")
        for rep in range(2):
            for rs in cfgs:
                ctx.set_tuning("upd_rs", rs)
                lk.double_gram_schmidt_step(B[k:k + p], B[:k], False, beta); ctx.sync()
                ctx.profile_reset(); ctx.profile_enable(True)
                t0 = time.perf_counter()
                for _ in range(5):
                    lk.double_gram_schmidt_step(B[k:k + p], B[:k], False, beta)
                ctx.sync(); dt = (time.perf_counter() - t0) / 5
                cnt, ms, by = ctx.profile_get("xhy_upd_mfma"); ctx.profile_enable(False)
                row["step_ms_%d" % rs] = round(dt * 1e3, 3); row["fused_pass_ms_%d" % rs] = round(ms / max(cnt, 1), 3)
        del B
        print(json.dumps(row), flush=True)
