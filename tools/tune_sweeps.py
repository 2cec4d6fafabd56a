#!/usr/bin/env python3
"""Sweep-kernel tuning on the GPU box: times the three DGS sweeps separately (HIP events on the
engine's stream) for a few (n, k) and tuning-knob settings.  Usage: python tools/tune_sweeps.py [f64|c128]"""
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "f64"
dtype = np.float64 if kind == "f64" else np.complex128
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else (40_000_000 if kind == "f64" else 20_000_000)
ks = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 32, 64, 128]
ctx = lk.Context(device=0)
kmax = max(ks)
B = lk.krylov_basis_gpu(n, kmax + 1, dtype, ctx)
for j in range(kmax + 1):
    B[j].rand(True, seed=100 + j)

def run(k, reps=5):
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(reps):
        lk.double_gram_schmidt_step(B[kmax], B[:k], if_chk_orthonormal=False)
    out = {}
    for tag in ("dgs_sweep1", "dgs_sweep2", "dgs_sweep3", "dgs_sweep*", "dgs"):
        c, ms, by = ctx.profile_get(tag)
        out[tag] = round(by / ms / 1e6, 1) if ms > 0 else 0.0     # GB/s
    ctx.profile_enable(False)
    return out

configs = [dict(grid_mult=g, prefetch=p, stream_update=s, update_grid_mult=u)
           for g, p, s, u in [(2, 0, 0, 4), (2, 1, 0, 4), (2, 1, 1, 2), (2, 1, 1, 4), (2, 1, 1, 8), (1, 1, 1, 4), (3, 1, 1, 4), (4, 1, 1, 4), (4, 0, 1, 4)]]
for k in ks:
    for cfg in configs:
        for key, v in cfg.items():
            ctx.set_tuning(key, v)
        run(k, 1)
        print(json.dumps({"kind": kind, "n": n, "k": k, **cfg, "GBps": run(k)}), flush=True)
