mkdir -p gpurun_out/r3i; D=gpurun_out/r3i
run() { tag=$1; shift; ( LK_DIST_BACKEND=gloo LK_FORCE_DEVICE=0 timeout 150 python bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline > $D/$tag.out 2> $D/$tag.err; echo "$tag rc=$? $(grep -c '^{' $D/$tag.out)" >> $D/summary.txt ); }
run g4_even --gpus 4 --rows 4000000 --kdim 32
run g4_odd --gpus 4 --rows 4000002 --kdim 32
run g3_odd --gpus 3 --rows 3000001 --kdim 20
run g4_k8 --gpus 4 --rows 4000002 --kdim 8
cat $D/summary.txt; tail -5 $D/g4_odd.err
