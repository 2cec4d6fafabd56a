#!/usr/bin/env bash
# scratch (round 4, third GPU trip)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/r4c; mkdir -p "$D"
cd "$R"
LK_TOL_REPORT=$D/tol.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 > "$D/pytest.log" 2>&1
echo "pytest rc $?" >> "$D/pytest.log"
for wr in 2 1; do
  python tools/bench_wide.py 1e7 f64 wide_regs=$wr > "$D/wide_f64_regs$wr.log" 2>&1
  python tools/bench_wide.py 5e6 c128 wide_regs=$wr > "$D/wide_c128_regs$wr.log" 2>&1
done
for u in 4 8 4 8; do
  python tools/bench_lincomb.py gemm_u=$u >> "$D/lincomb_u$u.log" 2>&1
done
python tools/bench_block.py 1e7 gemm_u=8 > "$D/block_u8.log" 2>&1
python tools/bench_block.py 1e7 gemm_u=4 > "$D/block_u4.log" 2>&1
tail -4 "$D/pytest.log"
