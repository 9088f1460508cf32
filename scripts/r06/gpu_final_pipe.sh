# the pipelined-run part of gpu_final.sh alone (PMC pass over the pipelined worker + the three pipelined traces), plus a trace of the
# PACED 80-scene operating point
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
p=gpurun_out/r06_pipe_pmc; mkdir -p $p
timeout 1200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --output-format csv -d $p -o pmc -- python3 bench.py --gpus 1 --worker --no-legs --steps 200 --warmup 50 --cpu-scenes 0 --no-roofline > $p/bench_stdout.log 2> $p/bench_stderr.log
grep '^{' $p/bench_stdout.log > $p/bench_under_profiler.json; cut -c1-200 $p/bench_under_profiler.json
python3 scripts/r06/pipeline_pmc_summary.py $p $p/bench_under_profiler.json > $p/pipeline_pmc_summary.json; head -30 $p/pipeline_pmc_summary.json
find $p -name "*.csv" -size +1M -delete
for sc in uniform beam 65536 paced; do
  out=gpurun_out/r06_pipe_$sc; mkdir -p $out
  if [ $sc = 65536 ]; then A="--steps 40 --warmup 8 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8"; elif [ $sc = beam ]; then A="--steps 200 --warmup 50 --scene beam"; elif [ $sc = paced ]; then A="--steps 200 --warmup 50 --headway-ms 5.55 --prefetch 2"; else A="--steps 200 --warmup 50"; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python3 bench.py $A --cpu-scenes 0 --no-roofline --no-legs --worker > $out/bench_stdout.log 2>&1
  grep '^{' $out/bench_stdout.log > $out/bench_under_profiler.json; cut -c1-200 $out/bench_under_profiler.json
  f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/pipeline_kernel_stats.csv
  t=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 scripts/r02/trace_summary.py $t > $out/trace_summary.txt; head -12 $out/trace_summary.txt
  python3 scripts/r06/trace_overlap.py $t > $out/trace_overlap.txt; head -14 $out/trace_overlap.txt
  rm -f $t
done
find gpurun_out -name "*.csv" -size +3M -delete
