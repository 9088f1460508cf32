# kernel trace of the PIPELINED run (16 main + 6 sampler streams, 24 hardware queues)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=24        # plain shell export: under rocprofv3 the runtime may start before bench.py sets it
out=gpurun_out/${1:-r02_prof_pipeline}; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python3 bench.py --steps 192 --warmup 48 --cpu-scenes 0 --no-roofline --no-legs > $out/bench_stdout.log 2>&1
grep '^{' $out/bench_stdout.log | cut -c1-400
f=$(find $out -name "*kernel_stats.csv" | head -1); echo $f; head -14 $f | cut -c1-220
t=$(find $out -name "*kernel_trace.csv" | head -1); ls -la $t
python3 scripts/r02/trace_summary.py $t > $out/trace_summary.txt; cat $out/trace_summary.txt | head -60
gzip -9 $t
