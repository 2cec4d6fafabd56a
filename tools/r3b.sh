mkdir -p gpurun_out/r3b; D=gpurun_out/r3b
(timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_sharded_emulation.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -30) > $D/tests.log
for kn in "gemm_grid_mult=4" "gemm_grid_mult=2" "gemm_grid_mult=1" "gemm_grid_mult=8" "gemm_store_policy=0" "gemm_store_policy=1" "gemm_store_policy=3" "gemm_mfma_min=100"; do
  echo "## $kn" >> $D/lincomb_knobs.log
  LK_LINCOMB_SHAPE=f64,10000000,64,32 python tools/bench_lincomb.py $kn 2>/dev/null | grep '^{' >> $D/lincomb_knobs.log
  LK_LINCOMB_SHAPE=c128,1000000,128,16 python tools/bench_lincomb.py $kn 2>/dev/null | grep '^{' >> $D/lincomb_knobs.log
done
python tools/explore_gl_parity.py 512 16 8 > $D/gl_512.log 2>&1
python tools/explore_gl_parity.py 100000 128 32 > $D/gl_1e5.log 2>&1
timeout 900 python tools/explore_gl_parity.py 1000000 128 64 > $D/gl_1e6.log 2>&1
python bench.py --operator dense --steps 2 --warmup 1 > $D/bench_dense.log 2> $D/bench_dense.err
python bench.py --operator lap5 --steps 5 --warmup 2 > $D/bench_lap5.log 2> $D/bench_lap5.err
python bench.py --operator csr --steps 5 --warmup 2 > $D/bench_csr.log 2> $D/bench_csr.err
python bench.py --no-cpu-baseline > $D/bench_diag.log 2> $D/bench_diag.err
cat $D/tests.log | tail -30
