#!/usr/bin/env bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/r4e; mkdir -p "$D"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "wide or sweep3 or knobs or 256" > "$D/pytest_wide.log" 2>&1
echo "rc $?" >> "$D/pytest_wide.log"
for s3 in 1 0; do
  python tools/bench_wide.py 1e7 f64 wide_s3=$s3 > "$D/wide_f64_s3$s3.log" 2>&1
  python tools/bench_wide.py 5e6 c128 wide_s3=$s3 > "$D/wide_c128_s3$s3.log" 2>&1
done
tail -5 "$D/pytest_wide.log"
