cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_tuning_knobs.py tests/test_gpu_lincomb.py tests/test_gpu_block_dgs.py -x -q 2>&1 | tail -5
for args in "gemm_roll=0" "gemm_roll=1"; do
  echo "== $args"
  python tools/bench_lincomb.py $args 2>&1 | grep "complex128\|float64" | cut -c1-200
  LK_LINCOMB_SHAPE=c128,5000000,128,64 python tools/bench_lincomb.py $args 2>&1 | grep complex | cut -c1-200
  LK_LINCOMB_SHAPE=c128,5000000,128,32 python tools/bench_lincomb.py $args 2>&1 | grep complex | cut -c1-200
  LK_LINCOMB_SHAPE=c128,5000000,64,16 python tools/bench_lincomb.py $args 2>&1 | grep complex | cut -c1-200
  python tools/bench_block.py 1e7 $args 2>&1 | tail -12 | cut -c1-250
done
