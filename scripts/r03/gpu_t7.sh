# after the final run: selective zero-fill of the pooled rows, cooperative launches bounded by the CU count, --gpus N > devices
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
python3 -m pytest tests/test_compact_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
python3 -m pytest tests/test_timed_path_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 120 python3 bench.py --gpus 2 --steps 20 --warmup 5 > /tmp/two.log 2>&1; echo "rc of --gpus 2 with one device: $?"; tail -1 /tmp/two.log
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2; do
python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform"
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam"
python3 bench.py $B --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 2>/dev/null | python3 -c "$show" "65536"
done
for sc in uniform beam; do
  out=gpurun_out/r03_pipe_$sc; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python3 bench.py --steps 192 --warmup 48 --cpu-scenes 0 --no-roofline --no-legs --scene $sc > $out/bench_stdout.log 2>&1
  grep '^{' $out/bench_stdout.log > $out/bench_under_profiler.json; cut -c1-160 $out/bench_under_profiler.json
  f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/pipeline_kernel_stats.csv
  t=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 scripts/r02/trace_summary.py $t > $out/trace_summary.txt; head -14 $out/trace_summary.txt
  rm -f $t
done
