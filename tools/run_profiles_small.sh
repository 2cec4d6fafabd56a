#!/usr/bin/env bash
# Runs ON THE GPU BOX: evidence for the launch-bound regime (round 6) -- kernel-trace statistics of small-n Arnoldi factorisations, HBM
# traffic of the single-launch Gram-Schmidt step against the three sweeps (PMC passes of their own), TLB counters for the footprint question.
#   gpurun -- 'bash tools/run_profiles_small.sh r6s'   ->  gpurun_out/r6s/*
set -u
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/$TAG; mkdir -p "$D"
cd /tmp; export TMPDIR=/tmp
P="python3 $R/tools/dgs_once.py"
pmc() { # name, counters, command...
  local name=$1 ctr=$2; shift 2
  if ! rocprofv3 --pmc $ctr --output-format csv -d "$D/$name" -o p -- "$@" > "$D/$name.log" 2>&1; then echo "rocprofv3 FAILED for $name" > "$D/pmc_$name.txt"; return 1; fi
  python3 "$R/tools/pmc_sum.py" "$D/$name" ALL > "$D/pmc_$name.txt" 2>&1
}
# -- 1. per-kernel time of small-n factorisations (single launch vs three sweeps)
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats_single" -o s -- python3 "$R/tools/arnoldi_once.py" 175000 64 5 > "$D/stats_single.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats_sweeps" -o s -- python3 "$R/tools/arnoldi_once.py" 175000 64 5 resident=0 > "$D/stats_sweeps.log" 2>&1
# -- 2. HBM traffic per launch: on-chip kernel, cache-resident kernel, three sweeps (FETCH_SIZE is reported at half on gfx950: pmc_sum doubles nothing in ALL mode)
for case in "175000 32" "300000 32" "300000 64"; do
  set -- $case; tag="n$1_k$2"
  pmc fetch_single_$tag FETCH_SIZE $P $1 $2 20
  pmc write_single_$tag WRITE_SIZE $P $1 $2 20
  pmc fetch_sweeps_$tag FETCH_SIZE $P $1 $2 20 resident=0
done
pmc fetch_cache_n300000_k32 FETCH_SIZE $P 300000 32 20 resident_onchip=0
# -- 3. address translation at two footprints (k = 128 three-sweep step): misses per request
for n in 30000000 100000000; do
  pmc tlb_n$n "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" $P $n 128 3
done
tail -n +1 "$D"/pmc_*.txt | cut -c1-200
