#!/usr/bin/env bash
# Runs in the build container after `gpurun -- 'bash tools/run_profiles.sh <dir> <commit>'`: turns gpurun_out/<dir>/ into the
# tracked files profiles/<tag>_*  (tools/make_profiles.py for the bench / kernel-stats / PMC records, the rest here).
#   bash tools/collect_profiles.sh r03r r03
set -euo pipefail
cd "$(dirname "$0")/.."
D=gpurun_out/$1; TAG=$2
C=$(cat "$D/commit.txt")
python tools/make_profiles.py "$D" "$TAG"
hdr="{\"commit\": \"$C\", \"box\": \"1 x MI355X (gpurun), tools/run_profiles.sh $1\"}"
j() { grep -h '^{' "$@" 2>/dev/null || true; }
(echo "$hdr"; j "$D/wide_f64.log" "$D/wide_c128.log") > "profiles/${TAG}_wide_dgs.jsonl"
(echo "$hdr"; j "$D/block.log") > "profiles/${TAG}_block_mfma.jsonl"
if [ -f "$D/block_wide.log" ]; then (echo "$hdr"; j "$D/block_wide.log") > "profiles/${TAG}_block_wide.jsonl"; fi
if [ -f "$D/gram.log" ]; then (echo "$hdr"; j "$D/gram.log") > "profiles/${TAG}_gram.jsonl"; fi
if [ -f "$D/shard_2.log" ]; then (echo "$hdr"; j "$D/shard_2.log" "$D/shard_4.log" "$D/shard_8.log") > "profiles/${TAG}_shard_lines_one_gpu.jsonl"; fi
if [ -f "$D/cfg5_8rank_one_gpu.log" ]; then (echo "$hdr"; j "$D/cfg5_2rank_one_gpu.log" "$D/cfg5_8rank_one_gpu.log") > "profiles/${TAG}_cfg5_ranks_sharing_one_gpu.jsonl"; fi
(echo "$hdr"; for op in dense lap5 csr; do j "$D/bench_$op.log"; done) > "profiles/${TAG}_operators.jsonl"
(echo "$hdr"; j "$D/blas1.log") > "profiles/${TAG}_blas1_n1e8.jsonl"
(echo "$hdr"; j "$D/per_object_arnoldi.log") > "profiles/${TAG}_per_object_arnoldi.jsonl"
if [ -f "$D/per_object_c.log" ]; then (echo "$hdr"; j "$D/per_object_c.log") > "profiles/${TAG}_per_object_c.jsonl"; fi
if [ -f "$D/small_arnoldi_f64.log" ]; then
  (echo "$hdr"; j "$D/scan_small_single.log"; j "$D/scan_small_sweeps.log") > "profiles/${TAG}_dgs_small_n_scan.jsonl"
  (echo "$hdr"; j "$D/small_arnoldi_f64.log" "$D/small_arnoldi_c128.log") > "profiles/${TAG}_small_arnoldi.jsonl"
  (echo "$hdr"; j "$D/resident_phases_f64.log") > "profiles/${TAG}_resident_phases_f64.jsonl"
  (echo "$hdr"; j "$D/block_arnoldi_f64.log") > "profiles/${TAG}_block_arnoldi.jsonl"
fi
(echo "$hdr"; j "$D/bench_default.log") > "profiles/${TAG}_bench_default_stdout.jsonl"
(echo "$hdr"; j "$D/cfg2.log") > "profiles/${TAG}_cfg2_f64_n1e7_m64.jsonl"
(echo "$hdr"; j "$D/configs.log") > "profiles/${TAG}_configs.jsonl"
if [ -f "$D/lincomb_scan_valu.log" ]; then
(echo "# VALU (gemm_mfma_min=100) vs MFMA (gemm_mfma_min=1) for narrow tall-skinny products; commit $C"; python - "$D" <<'PY'
import json, sys
D = sys.argv[1]
def load(f):
    out = {}
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l); out[(d['dtype'], d['k'], d['q'])] = (d['kernel_ms_per_call'], d['GBps_algorithmic(k+q cols)'], d['TFLOPs_fp64'])
    return out
v, m = load(D + '/lincomb_scan_valu.log'), load(D + '/lincomb_scan_mfma.log')
print("dtype k q | VALU ms GB/s | MFMA ms GB/s TFLOP/s | faster   (n = 1e7 real / 5e6 complex)")
for key in sorted(v):
    a, b = v[key], m.get(key, (0, 0, 0))
    print(key[0], key[1], key[2], '| %.3f %5.0f | %.3f %5.0f %5.1f |' % (a[0], a[1], b[0], b[1], b[2]), 'VALU' if a[0] < b[0] else 'MFMA')
PY
) > "profiles/${TAG}_lincomb_scan.txt"
fi
(echo "# rocprofv3 --pmc FETCH_SIZE (KB, x2 on gfx950) per kernel; commit $C"
 echo "## tools/bench_block_dgs.py 32 1  (n = 1e7 real, p = 32 against k = 64 and k = 128, 6 calls each: X 5.12 / 10.24 GB + Y 2.56 GB per pass => 122.88 GB per kernel = ONE pass each: three passes per block DGS)"
 if [ -d "$D/pmc_block_fetch" ]; then python tools/pmc_sum.py "$D/pmc_block_fetch" FETCH_SIZE; fi
 if [ -d "$D/pmc_block_wide_fetch" ]; then echo "## tools/bench_block_wide.py 4e6 panels_only (per kind: k = 256, p = 32 | 256, 4 | 192, 8 | 512, 32; 1 warm-up + 3 calls each; the panel schedule moves 4k - |last panel| columns of X per group of 32 columns of Y: (896 + 896 + 704 + 1920) columns x 4 calls x 32 MB (real, n = 4e6; complex n = 2e6: the same bytes) = 565 GB of X per kind over its coefficient products (panel_xhy_mfma*, panel_xhy_upd_mfma: A, C and the fused last panel of B) and updates (panel_gemm*: B, D), plus the columns of Y each pass touches)"; python tools/pmc_sum.py "$D/pmc_block_wide_fetch" FETCH_SIZE; grep -h "^{" "$D/pmc_block_wide_fetch.log" | cut -c1-260; fi
 if [ -d "$D/pmc_wide_fetch" ]; then
 echo "## tools/bench_wide.py 4e6 f64  (k = 64..640, 5 calls each; panel_sweep<f64, KC, 8, UPDATE, DOT, TWO, SC, G>: <32,..,1,1> = k 129..256 (sum of k+1 = 749 columns x 160 MB = 119.84 GB), <24,..,2,1> sweep 2 and <48,..,1,2> sweep 3 = k 257..384 (964 columns = 154.24 GB), <16,..,4,1> = k 512: (k + 1) columns of 32 MB per launch = ONE pass each)"
 python tools/pmc_sum.py "$D/pmc_wide_fetch" FETCH_SIZE; fi
 if [ -f profiles/${TAG}_pmc_lds_note.txt ]; then cat profiles/${TAG}_pmc_lds_note.txt; fi) > "profiles/${TAG}_pmc_wide_and_block.txt"
if [ -d "$D/dense" ]; then find "$D/dense" -name "*kernel_stats.csv" -exec cp {} "profiles/${TAG}_dense_n65536_kernel_stats.csv" \;; fi
if [ -d "$D/block_stats" ]; then find "$D/block_stats" -name "*kernel_stats.csv" -exec cp {} "profiles/${TAG}_block_dgs_kernel_stats.csv" \;; fi
if [ -f "$D/eigs_profile.log" ]; then (echo "# tools/profile_eigs_cycle.py 5 (configs[3] eigs cycle, cProfile of the calling thread); commit $C"; grep -v amdgpu.ids "$D/eigs_profile.log") > "profiles/${TAG}_eigs_cycle_profile.txt"; fi
if [ -f "$D/tol.txt" ]; then sort -u "$D/tol.txt" > "profiles/${TAG}_parity_margins.txt"; fi
(echo "$hdr"; j "$D/cfg4_untraced.log") > "profiles/${TAG}_cfg4_untraced.jsonl"
echo "profiles/${TAG}_* written from $D (commit $C)"
