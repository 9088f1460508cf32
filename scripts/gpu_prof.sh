set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r01p
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01p -o bench -- python3 bench.py --steps 20 --warmup 4 --streams 1 --no-graph --cpu-scenes 0 --no-roofline > gpurun_out/prof_r01p/bench_stdout.log 2>&1
ls -R gpurun_out/prof_r01p | head -20
f=$(find gpurun_out/prof_r01p -name "*kernel_stats.csv" | head -1)
head -40 $f
find gpurun_out/prof_r01p -name "*kernel_trace.csv" -delete
