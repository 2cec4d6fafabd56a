#!/usr/bin/env python3
"""Stress of the N > 1 entry on ONE GPU (tools, not product): from inside a process that holds a live HIP context and device
memory -- the situation of a pytest session -- launch `python bench.py --gpus P` (gloo standing in for RCCL, every rank on
device 0) for P in a list, `reps` times each, one attempt per launch.  Every launch runs under bench.py's own watchdog, so a
stall ends as rc != 0 with each rank's phase markers and stacks; those are written to the log in full.  One JSON line per launch.

  python tools/stress_multirank.py --procs 2,3,4,8 --reps 20 --log gpurun_out/r4a/stress.jsonl
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", default="2,3,4,8")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--log", default="gpurun_out/stress.jsonl")
    ap.add_argument("--watchdog", type=int, default=90)
    ap.add_argument("--parent-gb", type=float, default=4.0, help="device memory the launching process itself holds")
    ap.add_argument("--budget", type=float, default=1500.0, help="stop launching after this many seconds")
    args = ap.parse_args()
    import numpy as np
    import lightkrylov_amd as lk
    ctx = lk.Context(device=0)                      # the live parent context
    cols = 8
    X = lk.krylov_basis_gpu(int(args.parent_gb * 1e9 / 8 / cols), cols, np.float64, ctx)
    X[0].rand(True, seed=1)
    ctx.sync()
    os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LK_DIST_BACKEND="gloo", LK_FORCE_DEVICE="0", GLOO_SOCKET_IFNAME="lo",
               LK_BENCH_WATCHDOG=str(args.watchdog))
    t_start, bad = time.time(), 0
    with open(args.log, "a") as log:
        for rep in range(args.reps):
            for P in [int(p) for p in args.procs.split(",")]:
                if time.time() - t_start > args.budget:
                    break
                cmd = [sys.executable, "bench.py", "--gpus", str(P), "--rows", str(1000000 * P + 2), "--kdim", "32", "--steps", "1",
                       "--warmup", "0", "--no-cpu-baseline"]
                t0 = time.time()
                try:
                    out = subprocess.run(cmd, capture_output=True, text=True, timeout=args.watchdog * 3 + 120, cwd=ROOT, env=env)
                    rc, so, se = out.returncode, out.stdout, out.stderr
                except subprocess.TimeoutExpired as exc:
                    rc, so, se = -999, (exc.stdout or b"").decode(errors="replace"), (exc.stderr or b"").decode(errors="replace")
                dt = time.time() - t0
                lines = [ln for ln in so.splitlines() if ln.startswith("{")]
                rec = {"P": P, "rep": rep, "rc": rc, "seconds": round(dt, 1)}
                if rc == 0 and lines:
                    j = json.loads(lines[-1])
                    rec.update(H_fro=j["config"]["H_fro"], value=j["value"], n_gpus=j["n_gpus"])
                else:
                    bad += 1
                    rec["stderr"] = se[-20000:]
                    rec["stdout"] = so[-2000:]
                X[1].rand(True, seed=rep)           # the parent keeps using its context between launches
                ctx.sync()
                log.write(json.dumps(rec) + "\n")
                log.flush()
                print(json.dumps({k: v for k, v in rec.items() if k not in ("stderr", "stdout")}), flush=True)
    print(f"stress: {bad} bad launch(es)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
