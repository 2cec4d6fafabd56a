"""Interleaved A/B of several BUILDS of the engine in ONE process (process-to-process variation on one box is ~3 %,
box-to-box ~10 %: neither is usable for kernel effects of a few per cent).

  python tools/ab_builds.py [f64|c128] [--n ROWS] [--ks 32,128] [--tune key=val,...] name=path/to/lib.so ...

Every library is loaded side by side through ctypes (its own context and stream); the basis is allocated once by the
first library and wrapped (lk_basis_wrap) by the others, so all builds stream the SAME device memory.  Rounds alternate
between the builds; the median over 5 rounds of the per-sweep algorithmic GB/s (HIP events inside each library) is
printed per k.  Results are also cross-checked: every build must return the same DGS coefficients to 1e-12."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (one HIP runtime per process: torch's)
from lightkrylov_amd import _capi  # noqa: E402

args = sys.argv[1:]
kind, n, ks, libs, tune = "f64", None, (32, 128), {}, {}
i = 0
while i < len(args):
    a = args[i]
    if a in ("f64", "c128"):
        kind = a
    elif a == "--n":
        i += 1; n = int(args[i])
    elif a == "--ks":
        i += 1; ks = tuple(int(v) for v in args[i].split(","))
    elif a == "--tune":
        i += 1; tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args[i].split(",") if kv}
    elif "=" in a:
        name, path = a.split("=", 1); libs[name] = os.path.abspath(path)
    i += 1
dtype = _capi.LK_F64 if kind == "f64" else _capi.LK_C128
if n is None:
    n = 100_000_000 if kind == "f64" else 50_000_000
kmax = max(ks)


def chk(lib, rc):
    if rc != 0:
        raise RuntimeError(lib.lk_last_error().decode())


class Build:
    def __init__(self, name, path, share=None):
        self.name = name
        self.lib = C.CDLL(path)
        for fn, (res, at) in _capi.SIGNATURES.items():
            f = getattr(self.lib, fn); f.restype = res; f.argtypes = at
        self.ctx = C.c_void_p()
        chk(self.lib, self.lib.lk_init(0, None, C.byref(self.ctx)))
        for key, val in tune.items():
            chk(self.lib, self.lib.lk_set_tuning(self.ctx, key.encode(), val))
        self.B = C.c_void_p()
        if share is None:
            chk(self.lib, self.lib.lk_basis_create(self.ctx, dtype, n, kmax + 1, C.byref(self.B)))
            for j in range(kmax + 1):
                chk(self.lib, self.lib.lk_vec_rand(self.B, j, 100 + j, 0, 1))
        else:
            dt, nl, nc, ld, ptr = C.c_int(), C.c_int64(), C.c_int(), C.c_int64(), C.c_void_p()
            chk(share.lib, share.lib.lk_basis_info(share.B, C.byref(dt), C.byref(nl), C.byref(nc), C.byref(ld), C.byref(ptr)))
            chk(self.lib, self.lib.lk_basis_wrap(self.ctx, dtype, n, kmax + 1, ld.value, ptr, C.byref(self.B)))

    def run(self, k, reps=3):
        lib = self.lib
        chk(lib, lib.lk_profile_reset(self.ctx)); chk(lib, lib.lk_profile_enable(self.ctx, 1))
        h = np.zeros(2 * k); info = C.c_int()
        for _ in range(reps):
            chk(lib, lib.lk_dgs(self.B, k, self.B, kmax, h.ctypes.data_as(C.POINTER(C.c_double)), None, 0, C.byref(info)))
        out = []
        for tag in ("dgs_sweep1", "dgs_sweep2", "dgs_sweep3", "dgs_sweep*"):
            cnt, ms, by = C.c_int64(), C.c_double(), C.c_double()
            chk(lib, lib.lk_profile_get(self.ctx, tag.encode(), C.byref(cnt), C.byref(ms), C.byref(by)))
            out.append(round(by.value / ms.value / 1e6) if ms.value > 0 else 0)
        chk(lib, lib.lk_profile_enable(self.ctx, 0))
        return out

    def coefficients(self, k):
        """DGS of a fresh random column: h for the cross-check between builds."""
        lib = self.lib
        chk(lib, lib.lk_vec_rand(self.B, kmax, 999, 0, 1))
        h = np.zeros(2 * k); info = C.c_int()
        chk(lib, lib.lk_dgs(self.B, k, self.B, kmax, h.ctypes.data_as(C.POINTER(C.c_double)), None, 0, C.byref(info)))
        return h


builds, first = [], None
for name, path in libs.items():
    b = Build(name, path, first)
    first = first or b
    builds.append(b)

for k in ks:
    ref = None
    for b in builds:
        h = b.coefficients(k)
        if ref is None:
            ref = h
        else:
            assert np.abs(h - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300), (b.name, np.abs(h - ref).max())
    res = {b.name: [] for b in builds}
    for rnd in range(5):
        for b in (builds if rnd % 2 == 0 else builds[::-1]):
            if rnd == 0:
                b.run(k, 1)
            res[b.name].append(b.run(k))
    for b in builds:
        med = np.median(np.array(res[b.name]), axis=0).astype(int).tolist()
        print(json.dumps({"kind": kind, "n": n, "k": k, "build": b.name, "tune": tune,
                          "median_GBps[s1,s2,s3,all]": med}), flush=True)
