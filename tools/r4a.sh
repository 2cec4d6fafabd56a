#!/usr/bin/env bash
# scratch (round 4, first GPU trip): full GPU suite with the tolerance report, the N > 1 stress, baselines of the kernels to be tuned
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/r4a; mkdir -p "$D"
cd "$R"
LK_TOL_REPORT=$D/tol.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > "$D/pytest.log" 2>&1
echo "pytest rc $?" >> "$D/pytest.log"
timeout 1300 python tools/stress_multirank.py --procs 2,3,4,8 --reps 6 --watchdog 90 --budget 1000 --log "$D/stress.jsonl" > "$D/stress.log" 2>&1
python tools/bench_wide.py 1e7 f64 > "$D/wide_f64.log" 2>&1
python tools/bench_wide.py 5e6 c128 > "$D/wide_c128.log" 2>&1
python tools/bench_block.py 1e7 > "$D/block.log" 2>&1
tail -5 "$D/pytest.log"; tail -3 "$D/stress.log"
