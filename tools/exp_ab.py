"""Interleaved A/B of sweep variants in ONE process (rule: never compare across boxes/processes)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk
kind = sys.argv[1] if len(sys.argv) > 1 else "f64"
dtype = np.float64 if kind == "f64" else np.complex128
n = 40_000_000 if kind == "f64" else 20_000_000
ctx = lk.Context(device=0)
kmax = 128
B = lk.krylov_basis_gpu(n, kmax + 1, dtype, ctx)
for j in range(kmax + 1):
    B[j].rand(True, seed=100 + j)
variants = {
    "A_sweep3_barrier_two": dict(recompute_update=1, stream_two=0),
    "B_sweep3_stream_two": dict(recompute_update=1, stream_two=1),
}
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
for k in (16, 64, 128):
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
        print(json.dumps({"kind": kind, "k": k, "variant": name, "median_GBps[s1,s2,s3,all]": med}), flush=True)
