"""Sum rocprofv3 --pmc counters per kernel.
  python tools/pmc_sum.py <dir> FETCH_SIZE   (KB -> GB; FETCH_SIZE x2 on gfx950)
  python tools/pmc_sum.py <dir> ALL          (every counter collected, per kernel: launches, sum per launch)"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
ctr = sys.argv[2]
if ctr == "ALL":
    agg, names = {}, []
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void lk::", "")
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        a = agg.setdefault(k, {}).setdefault(c, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    top = sorted(agg.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))[:10]
    for k, d in top:
        n = max(v[0] for v in d.values())
        print(f"{k[:90]}  launches {n}")
        for c in names:
            if c in d:
                print(f"    {c:36s} {d[c][1] / d[c][0]:16.0f} per launch")
    sys.exit(0)
agg = {}
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != ctr:
        continue
    k = r["Kernel_Name"].split("(")[0]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (c, kb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k[:70]:70s} launches {c:4d}   {ctr} {kb * 1024 * (2 if ctr == 'FETCH_SIZE' else 1) / 1e9:8.2f} GB")
