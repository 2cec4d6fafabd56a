#!/usr/bin/env bash
# scratch (round 4, second GPU trip)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/r4b; mkdir -p "$D"
cd "$R"
LK_TOL_REPORT=$D/tol.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > "$D/pytest.log" 2>&1
echo "pytest rc $?" >> "$D/pytest.log"
# the 4-process stall: repeat it inside ONE pytest session, short watchdog, collectives traced
LK_TEST_REPEAT=8 LK_BENCH_WATCHDOG=45 timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_golden.py -m gpu -q -p no:cacheprovider -k "fixture or shard_the_metric_workload and 4" > "$D/p4.log" 2>&1
echo "p4 rc $?" >> "$D/p4.log"
for wr in 1 0; do
  python tools/bench_wide.py 1e7 f64 wide_regs=$wr > "$D/wide_f64_regs$wr.log" 2>&1
  python tools/bench_wide.py 5e6 c128 wide_regs=$wr > "$D/wide_c128_regs$wr.log" 2>&1
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/block_stats" -o block -- python3 "$R/tools/bench_block_dgs.py" 32 1 > "$D/block_stats.log" 2>&1
cd "$R"
tail -4 "$D/pytest.log"; tail -3 "$D/p4.log"
