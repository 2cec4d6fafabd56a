"""Phases of the fused pass of the block Gram-Schmidt (panel_xhy_upd_mfma: Y' = Y - X H1 stored, H2 = X^H Y') by switching parts of the kernel
off (tuning key upd_debug: 1 = no update MFMAs, 2 = no dot MFMAs, 4 = no global loads after the first tile, 8 = no store of Y'; results
are wrong, only the time means something).   python tools/bench_upd_phases.py [rows] [k] [p]"""
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
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
p = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ctx = lk.Context(device=0)
B = lk.krylov_basis_gpu(n, k, np.float64, ctx)
Y = lk.krylov_basis_gpu(n, p, np.float64, ctx)
for j in range(k):
    B[j].rand(True, seed=10 + j)
for dbg in (0, 1, 2, 3, 4, 8, 12, 15, 7, 11):
    for j in range(p):
        Y[j].rand(True, seed=500 + j)
    ctx.set_tuning("upd_debug", dbg)
    lk.double_gram_schmidt_step(Y, B, if_chk_orthonormal=False); ctx.sync()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3):
        lk.double_gram_schmidt_step(Y, B, if_chk_orthonormal=False)
    ctx.sync()
    out = {}
    for tag in ("xhy_mfma", "xhy_upd_mfma", "lincomb"):
        cnt, ms, by = ctx.profile_get(tag)
        out[tag + "_ms"] = round(ms / max(cnt, 1), 3)
    ctx.profile_enable(False)
    off = [name for bit, name in ((1, "update MFMAs"), (2, "dot MFMAs"), (4, "global loads"), (8, "store of Y'")) if dbg & bit]
    print(json.dumps({"n": n, "k": k, "p": p, "upd_debug": dbg, "switched_off": off, **out}), flush=True)
