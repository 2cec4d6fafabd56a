#!/usr/bin/env bash
# Runs ON THE GPU BOX (gpurun) at a round's FINAL kernel sources: every measurement the tracked profiles/<tag>_* files are made from.
#   gpurun -- 'bash tools/run_profiles_final.sh r5z <commit>'      then here:  bash tools/collect_profiles.sh r5z r05
# Order matters (round-4 review): the PMC passes run FIRST and profiles/pmc_traffic.json is regenerated on the box from them, so the
# bench lines taken afterwards -- the default run and the kernel-trace run that become tracked evidence -- carry `traffic` measured on
# exactly these kernel sources (no STALE marker in a tracked file).  Counters are collected in passes of their own (--pmc alone).
set -u
TAG=${1:?tag}; COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/$TAG
mkdir -p "$D"; echo "$COMMIT" > "$D/commit.txt"
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_hash())" > "$D/kernel_source_sha256.txt"
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
# -- 1. HBM traffic of the three DGS sweeps: the single-GPU workload and rank 0's row block of the 2 / 4 / 8-rank jobs (one record per n_local)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -o bench -- $B > "$D/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -o bench -- $B > "$D/pmc_write.log" 2>&1
for P in 2 4 8; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch_s$P" -o bench -- $B --shard-of $P > "$D/pmc_fetch_s$P.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write_s$P" -o bench -- $B --shard-of $P > "$D/pmc_write_s$P.log" 2>&1
done
(cd "$R" && python3 tools/make_profiles.py "$D" onbox > "$D/make_profiles_onbox.log" 2>&1)     # profiles/pmc_traffic.json of THIS build, on the box
# -- 2. the bench lines (now with `traffic`), the kernel trace, the shard lines, the 8-process line
python3 "$R/bench.py" > "$D/bench_default.log" 2> "$D/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o bench -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$D/stats.log" 2>&1
for P in 2 4 8; do python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --shard-of $P > "$D/shard_$P.log" 2>/dev/null; done
(cd "$R" && LK_DIST_BACKEND=gloo LK_FORCE_DEVICE=0 GLOO_SOCKET_IFNAME=lo python3 bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline > "$D/cfg5_8rank_one_gpu.log" 2> "$D/cfg5_8rank_one_gpu.err")
(cd "$R" && LK_DIST_BACKEND=gloo LK_FORCE_DEVICE=0 GLOO_SOCKET_IFNAME=lo python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > "$D/cfg5_2rank_one_gpu.log" 2> "$D/cfg5_2rank_one_gpu.err")
if [ "${LK_PROFILE_PARTS:-all}" = "metric" ]; then tail -c 600 "$D/bench_default.log"; exit 0; fi     # (re-take of parts 1-2 only)
# -- 3. the other kernels and configs
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/lincomb" -o lincomb -- python3 "$R/tools/bench_lincomb.py" > "$D/lincomb.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/cfg4" -o cfg4 -- python3 "$R/bench.py" --dtype c128 --rows 1000000 --steps 5 --warmup 2 --no-cpu-baseline > "$D/cfg4.log" 2>&1
python3 "$R/bench.py" --rows 10000000 --kdim 64 --steps 5 --warmup 2 --no-cpu-baseline > "$D/cfg2.log" 2>&1
python3 "$R/bench.py" --dtype c128 --rows 1000000 --steps 5 --warmup 2 --no-cpu-baseline > "$D/cfg4_untraced.log" 2>&1
python3 "$R/tools/bench_configs.py" > "$D/configs.log" 2>&1
python3 "$R/tools/bench_blas1.py" 1e8 2 > "$D/blas1.log" 2>&1
python3 "$R/tools/bench_block.py" 1e7 > "$D/block.log" 2>&1
python3 "$R/tools/bench_block_wide.py" 1e7 > "$D/block_wide.log" 2>&1
python3 "$R/tools/bench_gram.py" 1e7 > "$D/gram.log" 2>&1
python3 "$R/tools/bench_wide.py" 1e7 f64 > "$D/wide_f64.log" 2>&1
python3 "$R/tools/bench_wide.py" 5e6 c128 > "$D/wide_c128.log" 2>&1
python3 "$R/tools/bench_per_object_arnoldi.py" 1e7 64 > "$D/per_object_arnoldi.log" 2>&1
# the same per-object call sequence from a COMPILED host (what the Fortran plugin costs without an interpreter)
gcc -O2 -o /tmp/bench_per_object "$R/tools/bench_per_object.c" -I"$R/include" -L"$R/lightkrylov_amd" -llightkrylov_hip -lm -Wl,-rpath,"$R/lightkrylov_amd" && (/tmp/bench_per_object 10000000 64; /tmp/bench_per_object 1000000 128; /tmp/bench_per_object 100000000 32) > "$D/per_object_c.log" 2>&1
# round 6: the launch-bound regime (single-launch step) and the block factorisation
python3 "$R/tools/scan_dgs.py" f64 sizes=175000,300000,1000000 ks=8,32,64,128 > "$D/scan_small_single.log" 2>&1
python3 "$R/tools/scan_dgs.py" f64 resident=0 sizes=175000,300000,1000000 ks=8,32,64,128 > "$D/scan_small_sweeps.log" 2>&1
python3 "$R/tools/bench_small_arnoldi.py" f64 > "$D/small_arnoldi_f64.log" 2>&1
python3 "$R/tools/bench_small_arnoldi.py" c128 > "$D/small_arnoldi_c128.log" 2>&1
python3 "$R/tools/resident_phases.py" f64 sizes=1000,175000,300000,1000000 ks=1,8,32,64,128 > "$D/resident_phases_f64.log" 2>&1
python3 "$R/tools/bench_block_arnoldi.py" f64 > "$D/block_arnoldi_f64.log" 2>&1
for op in dense lap5 csr; do python3 "$R/bench.py" --operator $op --steps 3 --warmup 1 > "$D/bench_$op.log" 2> "$D/bench_$op.err"; done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_block_wide_fetch" -o block -- python3 "$R/tools/bench_block_wide.py" 4e6 panels_only > "$D/pmc_block_wide_fetch.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/block_stats" -o block -- python3 "$R/tools/bench_block_dgs.py" 32 1 > "$D/block_stats.log" 2>&1
python3 "$R/tools/profile_eigs_cycle.py" 5 > "$D/eigs_profile.log" 2>&1
# -- 4. the GPU suite
(cd "$R" && LK_TOL_REPORT="$D/tol.txt" timeout 2400 python3 -m pytest tests -m gpu -q -p no:cacheprovider > "$D/pytest.log" 2>&1; echo "pytest rc $?" >> "$D/pytest.log")
tail -c 600 "$D/bench_default.log"; tail -3 "$D/pytest.log"
