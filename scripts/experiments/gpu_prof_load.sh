cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_load
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_load -o bench -- python3 bench.py --steps 66 --warmup 22 --streams 22 --cpu-scenes 0 --no-roofline > gpurun_out/prof_load/bench_stdout.log 2>&1
tail -1 gpurun_out/prof_load/bench_stdout.log | cut -c1-200
f=$(find gpurun_out/prof_load -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-200
find gpurun_out/prof_load -name "*kernel_trace.csv" -delete
