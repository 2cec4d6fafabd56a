"""Interleaved A/B of engine tuning variants in ONE process (rule: never compare across boxes/processes).

  python tools/exp_ab.py [f64|c128] [--n ROWS] [--ks 16,64,128] name=key:val[,key:val] name2=...

Each variant is a set of lk_set_tuning keys; rounds alternate between the variants, the median over 4
rounds of the per-sweep algorithmic GB/s (HIP events inside the library) is printed per k."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

args = sys.argv[1:]
kind = "f64"
n = None
ks = (16, 64, 128)
variants = {}
i = 0
while i < len(args):
    a = args[i]
    if a in ("f64", "c128"):
        kind = a
    elif a == "--n":
        i += 1; n = int(args[i])
    elif a == "--ks":
        i += 1; ks = tuple(int(v) for v in args[i].split(","))
    elif "=" in a:
        name, cfg = a.split("=", 1)
        variants[name] = {kv.split(":")[0]: int(kv.split(":")[1]) for kv in cfg.split(",") if kv}
    i += 1
if not variants:
    variants = {"A_plain": dict(store_policy=0), "B_nt": dict(store_policy=1)}
dtype = np.float64 if kind == "f64" else np.complex128
if n is None:
    n = 40_000_000 if kind == "f64" else 20_000_000
ctx = lk.Context(device=0)
kmax = max(ks)
B = lk.krylov_basis_gpu(n, kmax + 1, dtype, ctx)
for j in range(kmax + 1):
    B[j].rand(True, seed=100 + j)
defaults = {}
for cfg in variants.values():
    for key in cfg:
        defaults.setdefault(key, None)

def run(k, reps=3):
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(reps):
        lk.double_gram_schmidt_step(B[kmax], B[:k], if_chk_orthonormal=False)
    out = []
    for tag in ("dgs_sweep1", "dgs_sweep2", "dgs_sweep3", "dgs_sweep*"):
        c, ms, by = ctx.profile_get(tag)
        out.append(round(by / ms / 1e6) if ms > 0 else 0)
    ctx.profile_enable(False)
    return out

for k in ks:
    res = {v: [] for v in variants}
    for rnd in range(4):
        for name, cfg in variants.items():
            for key, val in cfg.items():
                ctx.set_tuning(key, val)
            if rnd == 0:
                run(k, 1)
            res[name].append(run(k))
    for name in variants:
        med = np.median(np.array(res[name]), axis=0).astype(int).tolist()
        print(json.dumps({"kind": kind, "n": n, "k": k, "variant": name, "cfg": variants[name],
                          "median_GBps[s1,s2,s3,all]": med}), flush=True)
