# round 5, third measurement: GPU suite (new cooperative pre-pass, adaptive query cut), timing, pipelined traces with the overlap analysis
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_t3; mkdir -p $out
( timeout 2400 python3 -m pytest tests -m gpu -q -x ) > $out/pytest_all.log 2>&1; tail -6 $out/pytest_all.log | cut -c1-200
timeout 600 python3 tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -6
export GPU_MAX_HW_QUEUES=24
for i in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/bench_$i.log 2> $out/bench_$i.err
grep '^{' $out/bench_$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('uniform', d['value'], d['selfcheck'], d['latency']['ms_per_batch'], d['latency_b1']['ms_per_frame'], d['latency_under_load']['ms_p50_p99'])"
done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --scene beam > $out/bench_beam.log 2> $out/bench_beam.err
grep '^{' $out/bench_beam.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam', d['value'], d['selfcheck'])"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --leg-roofline --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 > $out/bench_65536.log 2> $out/bench_65536.err
grep '^{' $out/bench_65536.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('65536', d['value'], d['selfcheck'], d['latency_under_load']['ms_p50_p99'], d.get('index_kernels',{}).get('fps_us_per_round'))"
STEPS=4 NOPMC=1 bash scripts/r04/gpu_pmc.sh r05t3 > $out/pmc.log 2>&1; grep "bq_grid\|fps_fat\|cell_sort\|fps_seq" gpurun_out/pmc_r05t3/launches_of_one_pass.txt
for sc in uniform 65536; do
  o=$out/pipe_$sc; mkdir -p $o
  if [ $sc = 65536 ]; then A="--steps 40 --warmup 8 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8"; else A="--steps 192 --warmup 48"; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o -o pipe -- python3 bench.py $A --cpu-scenes 0 --no-roofline --no-legs --worker > $o/bench_stdout.log 2>&1
  grep '^{' $o/bench_stdout.log | cut -c1-160
  t=$(find $o -name "*kernel_trace.csv" | head -1)
  python3 scripts/r02/trace_summary.py $t > $o/trace_summary.txt; head -14 $o/trace_summary.txt
  python3 scripts/r05/trace_overlap.py $t > $o/trace_overlap.txt; cat $o/trace_overlap.txt
  rm -f $t
done
find $out -name "*.csv" -size +3M -delete
