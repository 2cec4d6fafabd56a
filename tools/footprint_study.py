"""Does the FOOTPRINT of the panel cost bandwidth?  (round-5 review, item 7: 6.84 TB/s at n = 3 10^7, k = 64 against 6.17 at n = 10^8, k = 128 in
one record, 0.815 of 8 TB/s for both a 1.25 10^7-row shard and the full problem in another.)  All panels are allocated ONCE and stay
allocated; the cases are measured INTERLEAVED, `rounds` times round-robin, with the GPU's clocks and power sampled between them
(rocm-smi), so that box-to-box and minute-to-minute drift cannot pose as a footprint effect.
  python tools/footprint_study.py [rounds]        ->  JSON lines: one per (round, case) + a summary per case"""
import json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightkrylov_amd as lk

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = lk.Context(device=0)
CASES = [(12_500_000, 128), (30_000_000, 64), (30_000_000, 128), (100_000_000, 64), (100_000_000, 128)]
panels = {}
for n in sorted({c[0] for c in CASES}):
    B = lk.krylov_basis_gpu(n, 129, np.float64, ctx)
    for j in range(129):
        B[j].rand(True, seed=100 + j)
    panels[n] = B


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "junction", "memory)"))}
        return keep
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


acc = {c: [] for c in CASES}
for r in range(rounds):
    for (n, k) in CASES:
        B = panels[n]
        lk.double_gram_schmidt_step(B[128], B[:k], False)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(3):
            lk.double_gram_schmidt_step(B[128], B[:k], False)
        c2, ms2, by2 = ctx.profile_get("dgs_sweep*")
        per = {t: ctx.profile_get(t) for t in ("dgs_sweep1", "dgs_sweep2", "dgs_sweep3")}
        ctx.profile_enable(False)
        tb = by2 / ms2 / 1e9
        acc[(n, k)].append(tb)
        print(json.dumps({"round": r, "n": n, "k": k, "sweeps_TBps": round(tb, 3),
                          "by_sweep_TBps": {t: round(v[2] / v[1] / 1e9, 3) for t, v in per.items()}, "smi": smi()}), flush=True)
for (n, k), v in acc.items():
    print(json.dumps({"summary": True, "n": n, "k": k, "panel_GB": round(n * (k + 1) * 8 / 1e9, 1), "median_TBps": round(float(np.median(v)), 3),
                      "min": round(min(v), 3), "max": round(max(v), 3), "rounds": len(v)}), flush=True)
