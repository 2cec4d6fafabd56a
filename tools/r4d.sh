#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/r4d; mkdir -p "$D"
cd "$R"
LK_TOL_REPORT=$D/tol.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > "$D/pytest.log" 2>&1
echo "pytest rc $?" >> "$D/pytest.log"
python tools/profile_eigs_cycle.py 5 > "$D/eigs_profile.log" 2>&1
python tools/bench_wide.py 1e7 f64 > "$D/wide_f64.log" 2>&1
python tools/bench_wide.py 5e6 c128 > "$D/wide_c128.log" 2>&1
tail -6 "$D/pytest.log"
