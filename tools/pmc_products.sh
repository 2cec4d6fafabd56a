#!/usr/bin/env bash
# Runs ON THE GPU BOX: HBM read traffic (FETCH_SIZE, counters only) of the tall-skinny product and of the Gram kernels, per kernel, against their algorithmic bytes.
#   gpurun -- 'bash tools/pmc_products.sh r5pmc'   ->  gpurun_out/r5pmc/pmc_{lincomb,gram}.txt
set -u
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/$TAG; mkdir -p "$D"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/lincomb" -o p -- python3 "$R/tools/bench_lincomb.py" > "$D/lincomb.log" 2>&1
python3 "$R/tools/pmc_sum.py" "$D/lincomb" FETCH_SIZE > "$D/pmc_lincomb.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/gram" -o p -- python3 "$R/tools/bench_gram.py" 1e7 > "$D/gram.log" 2>&1
python3 "$R/tools/pmc_sum.py" "$D/gram" FETCH_SIZE > "$D/pmc_gram.txt" 2>&1
grep -h "^{" "$D/lincomb.log" | cut -c1-200 > "$D/lincomb_lines.txt"
tail -n +1 "$D"/pmc_*.txt | cut -c1-200
