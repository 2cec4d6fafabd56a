#!/usr/bin/env python3
"""Turn one round's rocprofv3 outputs (gpurun_out/<dir>/...) into the tracked evidence under profiles/.

  python tools/make_profiles.py gpurun_out/r02p r02

Expects, under the given directory (written on the GPU box by the commands recorded in the JSON this script emits):
  stats/        rocprofv3 --kernel-trace --stats --output-format csv  -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  stats.log     stdout of that run (the bench JSON line)
  pmc_fetch/    rocprofv3 --pmc FETCH_SIZE --output-format csv        -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  pmc_write/    rocprofv3 --pmc WRITE_SIZE --output-format csv        -- (same)
  lincomb/, cfg4/ (optional)  kernel stats of tools/bench_lincomb.py and of bench.py --dtype c128 --rows 1000000
  commit.txt    git commit of the binary that ran
HBM traffic follows MI355X_MICROARCH.md (HBM section): separate --pmc passes; FETCH_SIZE/WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports 1/2 of a 16-B-per-lane streaming read -> doubled (calibrated on k_scal's known read); WRITE_SIZE exact."""
import csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
commit = open(os.path.join(src, "commit.txt")).read().strip() if os.path.exists(os.path.join(src, "commit.txt")) else "unknown"
# SHA-256 of the kernel sources the profiled binary was built from (written on the GPU box by tools/run_profiles.sh);
# bench.py refuses a traffic record whose hash is not that of the sources it sits beside
khash = (open(os.path.join(src, "kernel_source_sha256.txt")).read().strip()
         if os.path.exists(os.path.join(src, "kernel_source_sha256.txt")) else None)


def find(sub, suffix):
    hits = glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def bench_line(log):
    if not os.path.exists(log):
        return None
    lines = [ln for ln in open(log) if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def counter_sums(path, counter):
    agg = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


rec = {"commit": commit, "generated_by": "tools/make_profiles.py " + " ".join(sys.argv[1:])}
st = find("stats", "kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(out, f"{tag}_bench_n1e8_m128_kernel_stats.csv"))
b = bench_line(os.path.join(src, "stats.log"))
if b:
    b["profiled_with"] = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
    b["commit"] = commit
    json.dump(b, open(os.path.join(out, f"{tag}_bench_n1e8_m128.json"), "w"), indent=1)
for sub, name in (("lincomb", "lincomb"), ("cfg4", "cfg4_c128_n1e6_m128")):
    f = find(sub, "kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(out, f"{tag}_{name}_kernel_stats.csv"))
    lg = os.path.join(src, sub + ".log")
    if os.path.exists(lg):
        lines = [ln for ln in open(lg) if ln.startswith("{")]
        if lines:
            open(os.path.join(out, f"{tag}_{name}.jsonl"), "w").write(json.dumps({"commit": commit}) + "\n" + "".join(lines))

def traffic_record(sub_fetch, sub_write, shard_of):
    """One PMC traffic record (per n_local) from a FETCH_SIZE pass and a WRITE_SIZE pass of the same bench.py command."""
    ff, fw = find(sub_fetch, "counter_collection.csv"), find(sub_write, "counter_collection.csv")
    if not (ff and fw):
        return None, None
    fetch, write = counter_sums(ff, "FETCH_SIZE"), counter_sums(fw, "WRITE_SIZE")
    pb = bench_line(os.path.join(src, sub_fetch + ".log")) or {}
    n_local = pb.get("config", {}).get("n_local", 100_000_000)
    m = pb.get("config", {}).get("m", 128)
    is_sweep = lambda k: "panel_sweep" in k or "panel_dot_cw" in k          # sweep 1 is panel_dot_cw, sweeps 2 and 3 panel_sweep
    sweeps = [k for k in fetch if is_sweep(k)]
    launches = sum(fetch[k][0] for k in sweeps)
    # calibration of the gfx950 FETCH_SIZE halving on a kernel with a known read: k_scal reads n doubles
    scal = [k for k in fetch if "k_scal" in k]
    calib = None
    if scal:
        cnt, kb = fetch[scal[0]]
        calib = (8.0 * n_local) / (kb / cnt * 1024.0)
    fetch_b = sum(fetch[k][1] for k in sweeps) * 1024.0 * 2.0 / max(launches, 1)
    write_b = sum(write[k][1] for k in write if is_sweep(k)) * 1024.0 / max(launches, 1)
    alg = 8.0 * n_local * sum(3 * k + 5 for k in range(1, m + 1)) / (3.0 * m)
    must = 8.0 * n_local * sum(3 * k + 4 for k in range(1, m + 1)) / (3.0 * m)
    flags = " --steps 1 --warmup 0 --no-cpu-baseline" + (f" --shard-of {shard_of}" if shard_of > 1 else "")
    name = f"{tag}_pmc_n{n_local:.3g}_m{m}.json".replace("+0", "").replace("+", "")
    pm = {
        "commit": commit, "kernel_source_sha256": khash, "n_local": n_local, "m": m, "dtype": "f64",
        "shard": (f"rank 0's row block of a {shard_of}-rank job (n_global = {pb.get('config', {}).get('n_global')}), alone on one GPU: bench.py --shard-of {shard_of}"
                  if shard_of > 1 else "the single-GPU workload"),
        "command_fetch": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py" + flags,
        "command_write": "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python3 bench.py" + flags,
        "method": "separate --pmc passes; KB units; FETCH_SIZE doubled (gfx950 reports 1/2 of a 16 B/lane streaming read, MI355X_MICROARCH.md); "
                  "WRITE_SIZE exact; per launch = sum over the three DGS sweep kernels (panel_dot_cw, panel_sweep x2) / their launch count",
        "fetch_calibration_on_k_scal(expected 2.0)": calib,
        "sweep_launches": launches,
        "fetch_bytes_per_launch_corrected": fetch_b, "write_bytes_per_launch": write_b,
        "hbm_bytes_per_launch": fetch_b + write_b,
        "algorithmic_bytes_per_launch": alg, "bytes_per_launch_this_schedule_must_move": must,
        "traffic_over_algorithmic": (fetch_b + write_b) / alg, "traffic_over_must_move": (fetch_b + write_b) / must,
        "per_kernel": [{"kernel": k[:80], "counter": c, "launches": v[0], "sum_KB": v[1]}
                       for c, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)) for k, v in sorted(d.items())],
    }
    json.dump(pm, open(os.path.join(out, name), "w"), indent=1)
    short = {"n_local": n_local, "m": m, "dtype": "f64", "hbm_bytes_per_launch": fetch_b + write_b,
             "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch_b + write_b) / alg,
             "commit": commit, "kernel_source_sha256": khash, "source": f"profiles/{name}", "method": pm["method"]}
    return short, calib


# one record per rows-per-rank: the single-GPU workload and rank 0's block of the 2 / 4 / 8-rank jobs (bench.py --shard-of P)
records = []
for shard_of, suffix in ((1, ""), (2, "_s2"), (4, "_s4"), (8, "_s8")):
    short, calib = traffic_record("pmc_fetch" + suffix, "pmc_write" + suffix, shard_of)
    if short:
        records.append(short)
        rec.setdefault("pmc", []).append({"n_local": short["n_local"], "traffic_over_algorithmic": short["traffic_over_algorithmic"], "calibration": calib})
if records:
    json.dump({"records": records,
               "note": "one record per n_local (rows per rank): the sweep kernels of a rank see only their row block, so one GPU measures every shard "
                       "size (bench.py --shard-of P under rocprofv3 --pmc, tools/run_profiles.sh); written by tools/make_profiles.py"},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
# the scaling model bench.py prints next to `value`: the single-GPU factorisation and, measured on ONE GPU, the local work of
# a rank of the 2 / 4 / 8-rank jobs (bench.py --shard-of P: rank 0's row block alone, no collectives); the collective cost stays an assumption
b1 = bench_line(os.path.join(src, "bench_default.log"))
if b1 and b1.get("config", {}).get("operator") == "diag" and tag != "onbox":
    shard = {}
    for P in (2, 4, 8):
        bs = bench_line(os.path.join(src, f"shard_{P}.log"))
        if bs and bs.get("shard_emulation", {}).get("of_ranks") == P:
            shard[str(P)] = bs["ms_per_step"]
    json.dump({
        "workload": {"operator": "diag", "dtype": b1["dtype"], "n_global": b1["config"]["n_global"], "m": b1["config"]["m"]},
        "single_gpu_ms_per_factorisation": b1["ms_per_step"],
        "shard_ms_per_factorisation": shard,
        "source": f"tools/run_profiles_final.sh at commit {commit} (1 x MI355X): python bench.py ({b1['steps']} x {b1['ms_per_step']:.1f} ms, {b1['value']:.2f} Arnoldi it/s); "
                  "shards: python bench.py --steps 3 --warmup 1 --shard-of P",
        "allreduce_us_assumed": 30.0,
        "finish_partials_us": 4.0,
        "note": "DESIGN.md section 6: a factorisation on N ranks costs the local work of a rank -- measured on one GPU for N = 2, 4, 8 (a rank's sweeps see only its "
                "row block), T1 / N otherwise -- plus m*3*(allreduce + finish); the all-reduce latency is an ASSUMPTION until an 8-GPU node has run "
                "(<= 129 doubles, ring over xGMI, LL protocol: launch + N-1 hops)",
    }, open(os.path.join(out, "scaling_model.json"), "w"), indent=1)
print(json.dumps(rec))
