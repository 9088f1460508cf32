set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r01r}
mkdir -p gpurun_out/prof_$TAG
python scripts/gpu_linear_breakdown.py 2>/dev/null | tee gpurun_out/prof_$TAG/breakdown.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python3 bench.py --steps 20 --warmup 4 --streams 1 --no-graph --cpu-scenes 0 --no-roofline > gpurun_out/prof_$TAG/bench_stdout.log 2>&1
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-220
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
