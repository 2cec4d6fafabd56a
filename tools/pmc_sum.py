"""Sum a rocprofv3 --pmc counter per kernel: python tools/pmc_sum.py <dir> FETCH_SIZE   (KB -> GB; FETCH_SIZE x2 on gfx950)"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
ctr = sys.argv[2]
agg = {}
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != ctr:
        continue
    k = r["Kernel_Name"].split("(")[0]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (c, kb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"{k[:70]:70s} launches {c:4d}   {ctr} {kb * 1024 * (2 if ctr == 'FETCH_SIZE' else 1) / 1e9:8.2f} GB")
