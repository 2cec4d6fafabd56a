#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/r4f; mkdir -p "$D"
cd "$R"
for rep in 1 2; do
  for kc in 0 64 32 1; do
    python bench.py --steps 4 --warmup 1 --no-cpu-baseline --tune kc32=$kc > "$D/bench_kc${kc}_$rep.log" 2> "$D/bench_kc${kc}_$rep.err"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4f/bench_kc*_?.log')):
    for l in open(f):
        if l.startswith('{'):
            j=json.loads(l); r=j['roofline']
            print(f.split('/')[-1], round(j['value'],2), round(r['frac'],4), {k:round(v['frac'],4) for k,v in r['per_sweep'].items()})
PY
