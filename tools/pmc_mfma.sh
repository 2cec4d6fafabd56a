#!/usr/bin/env bash
# Runs ON THE GPU BOX: SQ counters of the matrix-core kernels (how busy is the MFMA pipe, where do the waves wait) -- counters only, no trace.
#   gpurun -- 'bash tools/pmc_mfma.sh r5e'   ->  gpurun_out/r5e/pmc_mfma_*.txt
set -u
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/$TAG; mkdir -p "$D"
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|GRBM_GUI_ACTIVE\|SQ_LDS_[A-Z_]*" | sort -u > "$D/counters_available.txt"
# at most FOUR SQ counters per pass (round-5 advisor: eight can exceed the SQ's counter slots per pass, and a partial collection went unnoticed)
C1a="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
C1b="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64"
C2a="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C2b="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
run() { # name, counters, command...
  local name=$1 ctr=$2; shift 2
  if ! rocprofv3 --pmc $ctr --output-format csv -d "$D/$name" -o p -- "$@" > "$D/$name.log" 2>&1; then
    echo "pmc_mfma: rocprofv3 FAILED for $name (see $D/$name.log): no summary written" | tee "$D/pmc_mfma_$name.txt"; return 1
  fi
  if [ -z "$(find "$D/$name" -name '*counter_collection.csv' -print -quit)" ]; then
    echo "pmc_mfma: no counter_collection.csv for $name: no summary written" | tee "$D/pmc_mfma_$name.txt"; return 1
  fi
  python3 "$R/tools/pmc_sum.py" "$D/$name" ALL > "$D/pmc_mfma_$name.txt" 2>&1
}
for pass in 1a:"$C1a" 1b:"$C1b" 2a:"$C2a" 2b:"$C2b"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  run gram_old_$tag "$ctr" python3 "$R/tools/bench_gram.py" 1e7 gram_rs=0
  run gram_new_$tag "$ctr" python3 "$R/tools/bench_gram.py" 1e7 gram_rs=1
  run block_$tag "$ctr" python3 "$R/tools/bench_block_dgs.py" 32 1
  run lincomb_$tag "$ctr" python3 "$R/tools/bench_lincomb.py"
done
tail -n +1 "$D"/pmc_mfma_*.txt | cut -c1-220
