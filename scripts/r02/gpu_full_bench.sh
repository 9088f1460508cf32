cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_full
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r02_full/bench_20.log 2>&1
grep '^{' gpurun_out/r02_full/bench_20.log | cut -c1-300; tail -4 gpurun_out/r02_full/bench_20.log
