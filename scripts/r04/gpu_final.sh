# final artifacts of round 4: the driver's command (every leg), PMC + kernel stats on the three workloads (benchmark scenes,
# ray-cast scenes, 65536-point scenes), pipelined traces, --gpus 2 dry run on the one GPU of the box (gloo barrier)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
mkdir -p gpurun_out/r04_final
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04_final/bench_20.log 2> gpurun_out/r04_final/bench_20.err
grep '^{' gpurun_out/r04_final/bench_20.log | cut -c1-200; tail -4 gpurun_out/r04_final/bench_20.err
bash scripts/r04/gpu_pmc_all.sh z
for sc in uniform beam; do
  out=gpurun_out/r04_pipe_$sc; mkdir -p $out
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python3 bench.py --steps 192 --warmup 48 --cpu-scenes 0 --no-roofline --no-legs --scene $sc > $out/bench_stdout.log 2>&1
  grep '^{' $out/bench_stdout.log > $out/bench_under_profiler.json; cut -c1-200 $out/bench_under_profiler.json
  f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/pipeline_kernel_stats.csv
  t=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 scripts/r02/trace_summary.py $t > $out/trace_summary.txt; head -12 $out/trace_summary.txt
  rm -f $t
done
out=gpurun_out/r04_pipe_65536; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python3 bench.py --steps 40 --warmup 8 --cpu-scenes 0 --no-roofline --no-legs --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 > $out/bench_stdout.log 2>&1
grep '^{' $out/bench_stdout.log > $out/bench_under_profiler.json; cut -c1-200 $out/bench_under_profiler.json
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/pipeline_kernel_stats.csv; head -8 $f | cut -c1-150
find $out -name "*kernel_trace.csv" -delete
DET6D_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r04_final/bench_2ranks.log 2> gpurun_out/r04_final/bench_2ranks.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r04_final/bench_2ranks.log'):
    if l.startswith('{'):
        d = json.loads(l); print('2 ranks on one GPU (gloo dry run):', d['n_gpus'], d['value'], d['per_rank_scenes_per_s'], d['selfcheck'], d['ranks_seen'])
PY
