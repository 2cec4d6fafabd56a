#!/usr/bin/env bash
# Runs ON THE GPU BOX after kernel sources changed late in a round: the legs of tools/run_profiles.sh that must describe the FINAL
# sources -- bench default + kernel stats + the two PMC passes (profiles/pmc_traffic.json carries the kernel-source hash) -- plus the
# legs the late change touches (complex three-product kernels: lincomb, block) and the full GPU suite.
#   gpurun -- 'bash tools/run_profiles_final.sh r4q <commit>'   then: python tools/make_profiles.py gpurun_out/r4q r04
set -u
TAG=${1:?tag}; COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/$TAG
mkdir -p "$D"; echo "$COMMIT" > "$D/commit.txt"
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_hash())" > "$D/kernel_source_sha256.txt"
cd /tmp; export TMPDIR=/tmp
python3 "$R/bench.py" > "$D/bench_default.log" 2> "$D/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o bench -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$D/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -o bench -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$D/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -o bench -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$D/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/lincomb" -o lincomb -- python3 "$R/tools/bench_lincomb.py" > "$D/lincomb.log" 2>&1
for m in 1 0 1 0; do python3 "$R/tools/bench_lincomb.py" gemm_3m=$m 2>/dev/null | grep complex | sed "s/^{/{\"gemm_3m\": $m, /" >> "$D/complex_3m.log"; done
for m in 1 0 1 0; do python3 "$R/tools/bench_block.py" 1e7 gemm_3m=$m 2>/dev/null | grep complex | sed "s/^{/{\"gemm_3m\": $m, /" >> "$D/complex_3m.log"; done
python3 "$R/tools/bench_block.py" 1e7 > "$D/block.log" 2>&1
python3 "$R/tools/bench_configs.py" > "$D/configs.log" 2>&1
(cd "$R" && LK_TOL_REPORT="$D/tol.txt" timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider > "$D/pytest.log" 2>&1; echo "pytest rc $?" >> "$D/pytest.log")
tail -c 400 "$D/bench_default.log"; tail -3 "$D/pytest.log"
