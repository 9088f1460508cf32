# final artifacts of round 5: the whole GPU suite, the driver's command (every leg), PMC + kernel stats on the three workloads, the
# PMC pass over the PIPELINED worker (steady-state slice), pipelined traces with the overlap analysis, --gpus 2 dry run
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > gpurun_out/r05_final/pytest.log 2>&1; tail -4 gpurun_out/r05_final/pytest.log | cut -c1-200
export GPU_MAX_HW_QUEUES=24
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_final/bench_20.log 2> gpurun_out/r05_final/bench_20.err
grep '^{' gpurun_out/r05_final/bench_20.log | cut -c1-200; tail -4 gpurun_out/r05_final/bench_20.err
bash scripts/r05/gpu_pmc_all.sh z
p=gpurun_out/r05_pipe_pmc; mkdir -p $p
timeout 1200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --output-format csv -d $p -o pmc -- python3 bench.py --gpus 1 --worker --no-legs --steps 200 --warmup 50 --cpu-scenes 0 --no-roofline > $p/bench_stdout.log 2> $p/bench_stderr.log
grep '^{' $p/bench_stdout.log > $p/bench_under_profiler.json; cut -c1-200 $p/bench_under_profiler.json
python3 scripts/r05/pipeline_pmc_summary.py $p $p/bench_under_profiler.json > $p/pipeline_pmc_summary.json; head -30 $p/pipeline_pmc_summary.json
find $p -name "*.csv" -size +1M -delete
for sc in uniform beam 65536; do
  out=gpurun_out/r05_pipe_$sc; mkdir -p $out
  if [ $sc = 65536 ]; then A="--steps 40 --warmup 8 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8"; elif [ $sc = beam ]; then A="--steps 200 --warmup 50 --scene beam"; else A="--steps 200 --warmup 50"; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python3 bench.py $A --cpu-scenes 0 --no-roofline --no-legs --worker > $out/bench_stdout.log 2>&1
  grep '^{' $out/bench_stdout.log > $out/bench_under_profiler.json; cut -c1-200 $out/bench_under_profiler.json
  f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/pipeline_kernel_stats.csv
  t=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 scripts/r02/trace_summary.py $t > $out/trace_summary.txt; head -12 $out/trace_summary.txt
  python3 scripts/r05/trace_overlap.py $t > $out/trace_overlap.txt; head -6 $out/trace_overlap.txt
  rm -f $t
done
DET6D_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-legs > gpurun_out/r05_final/bench_2ranks.log 2> gpurun_out/r05_final/bench_2ranks.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r05_final/bench_2ranks.log'):
    if l.startswith('{'):
        d = json.loads(l); print('2 ranks on one GPU (gloo dry run):', d['n_gpus'], d['value'], d['per_rank_scenes_per_s'], d['selfcheck'], d['ranks_seen'])
PY
find gpurun_out -name "*.csv" -size +3M -delete
