# last refresh of the bench artifact on the final build (the driver's command, every leg) + the whole GPU suite
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
( time python3 -m pytest tests -x -q -m gpu ) 2>&1 | tail -5
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
mkdir -p gpurun_out/r03_final
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r03_final/bench_20.log 2> gpurun_out/r03_final/bench_20.err
grep '^{' gpurun_out/r03_final/bench_20.log | cut -c1-160; tail -4 gpurun_out/r03_final/bench_20.err
