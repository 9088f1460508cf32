# round 5: the lane-per-centre grid query — parity (all ball-query tests + the model-level ones), then same-box timing and the bench worker
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_bq; mkdir -p $out
( timeout 1500 python3 -m pytest tests/test_ball_query_shapes_gpu.py tests/test_ops_gpu.py tests/test_compact_gpu.py tests/test_model_gpu.py -m gpu -x -q ) > $out/pytest.log 2>&1; tail -15 $out/pytest.log
export GPU_MAX_HW_QUEUES=24
for i in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/bench_$i.log 2> $out/bench_$i.err
grep '^{' $out/bench_$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('uniform', d['value'], d['selfcheck'], d['latency'], d.get('latency_b1'))"
done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --scene beam > $out/bench_beam.log 2> $out/bench_beam.err
grep '^{' $out/bench_beam.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam', d['value'], d['selfcheck'])"
STEPS=4 NOPMC=1 bash scripts/r04/gpu_pmc.sh r05bq > $out/pmc.log 2>&1; grep "bq_grid\|compact" gpurun_out/pmc_r05bq/launches_of_one_pass.txt
