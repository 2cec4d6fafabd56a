#!/usr/bin/env bash
# Runs ON THE GPU BOX (gpurun): the rocprofv3 passes tools/make_profiles.py turns into profiles/<tag>_*.
#   gpurun -- 'bash tools/run_profiles.sh r02x <commit>'      then here:  python tools/make_profiles.py gpurun_out/r02x r02
# Counters are collected in passes of their own (--pmc alone), never together with a trace.
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
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/cfg4" -o cfg4 -- python3 "$R/bench.py" --dtype c128 --rows 1000000 --steps 5 --warmup 2 --no-cpu-baseline > "$D/cfg4.log" 2>&1
python3 "$R/bench.py" --rows 10000000 --kdim 64 --steps 5 --warmup 2 --no-cpu-baseline > "$D/cfg2.log" 2>&1
python3 "$R/tools/bench_configs.py" > "$D/configs.log" 2>&1
python3 "$R/tools/bench_blas1.py" 1e8 2 > "$D/blas1.log" 2>&1
python3 "$R/tools/bench_per_object_arnoldi.py" 1e7 64 > "$D/per_object_arnoldi.log" 2>&1
python3 "$R/tools/bench_block.py" 1e7 > "$D/block.log" 2>&1
gcc -O2 -o /tmp/bench_per_object "$R/tools/bench_per_object.c" -I"$R/include" -L"$R/lightkrylov_amd" -llightkrylov_hip -lm -Wl,-rpath,"$R/lightkrylov_amd" && (/tmp/bench_per_object 10000000 64; /tmp/bench_per_object 1000000 128; /tmp/bench_per_object 100000000 32) > "$D/per_object_c.log" 2>&1
python3 "$R/tools/bench_wide.py" 1e7 f64 > "$D/wide_f64.log" 2>&1
python3 "$R/tools/bench_wide.py" 5e6 c128 > "$D/wide_c128.log" 2>&1
LK_LINCOMB_SCAN=1 python3 "$R/tools/bench_lincomb.py" gemm_mfma_min=100 > "$D/lincomb_scan_valu.log" 2>&1
LK_LINCOMB_SCAN=1 python3 "$R/tools/bench_lincomb.py" gemm_mfma_min=1 > "$D/lincomb_scan_mfma.log" 2>&1
for op in dense lap5 csr; do python3 "$R/bench.py" --operator $op --steps 3 --warmup 1 > "$D/bench_$op.log" 2> "$D/bench_$op.err"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/dense" -o dense -- python3 "$R/bench.py" --operator dense --steps 1 --warmup 1 --no-cpu-baseline > "$D/dense_traced.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_wide_fetch" -o wide -- python3 "$R/tools/bench_wide.py" 4e6 f64 > "$D/pmc_wide_fetch.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_block_fetch" -o block -- python3 "$R/tools/bench_block_dgs.py" 32 1 > "$D/pmc_block_fetch.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/block_stats" -o block -- python3 "$R/tools/bench_block_dgs.py" 32 1 > "$D/block_stats.log" 2>&1
python3 "$R/tools/profile_eigs_cycle.py" 5 > "$D/eigs_profile.log" 2>&1
python3 "$R/bench.py" --dtype c128 --rows 1000000 --steps 5 --warmup 2 --no-cpu-baseline > "$D/cfg4_untraced.log" 2>&1
(cd "$R" && LK_TOL_REPORT="$D/tol.txt" timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider > "$D/pytest.log" 2>&1; echo "pytest rc $?" >> "$D/pytest.log")
tail -c 600 "$D/bench_default.log"; tail -3 "$D/pytest.log"
