"""Gram matrix X^H X on the matrix cores (lk_gram): ms per call, TFLOP/s on the flops of the upper 16 x 16 tiles, GB/s on one pass over X.
  python tools/bench_gram.py [rows] [KEY=INT ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
from lightkrylov_amd import _capi
# the keys that switch parts of a kernel off exist only in the phase-timing build (make -C lightkrylov_amd/csrc diagnostics)
_diag = os.path.join(os.path.dirname(_capi.LIB_PATH), "liblightkrylov_hip_diag.so")
if os.path.exists(_diag):
    _capi.LIB_PATH = _diag

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ctx = lk.Context(device=0)
tune = {}
for kv in sys.argv[2:]:
    key, val = kv.split("=")
    ctx.set_tuning(key, int(val)); tune[key] = int(val)
for dtype in (np.float64, np.complex128):
    nn = n if dtype == np.float64 else n // 2
    s = np.dtype(dtype).itemsize
    for k in (128, 96, 64, 48):
        B = lk.krylov_basis_gpu(nn, k, dtype, ctx)
        for j in range(k):
            B[j].rand(True, seed=10 + j)
        for _ in range(10):                       # (warm-up long enough for the clocks: the first few calls after an idle stretch run 10-15 % slower)
            lk.Gram(B)
        ctx.sync()
        ctx.profile_reset(); ctx.profile_enable(True)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            G = lk.Gram(B)
        ctx.sync()
        wall = (time.perf_counter() - t0) / reps
        cnt, ms, by = ctx.profile_get("xhy_mfma"); ctx.profile_enable(False)
        KP = (k + 15) // 16
        mfma_per_row = KP * (KP + 1) / 2 * (3 if s == 16 else 1) / 4.0        # 16x16x4 MFMAs per (complex) row: upper tiles, 3 products per complex one
        flop = mfma_per_row * 2048 * nn
        print(json.dumps({"dtype": np.dtype(dtype).name, "n": nn, "k": k, **tune, "kernel_ms": round(ms / cnt, 3), "wall_ms": round(wall * 1e3, 3),
                          "TFLOPs_on_upper_tiles": round(flop / (ms / cnt) / 1e9, 1), "GBps_one_pass_over_X": round(nn * s * k / (ms / cnt) / 1e6, 0),
                          "orth_err": float(np.abs(G - np.eye(k)).max()) if False else None}), flush=True)
        del B
