cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t11; mkdir -p $out
( timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_ball_query_shapes_gpu.py -m gpu -q -x ) > $out/pytest.log 2>&1; tail -4 $out/pytest.log | cut -c1-200
run() { tag=$1; shift
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], 'b8', d['latency']['ms_per_batch'], 'b1', d['latency_b1']['ms_per_frame'])" || tail -3 $out/b_$tag.err; }
run ship
run ship_beam --scene beam
run ship_65536 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8
run ship_65536b --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8
STEPS=4 NOPMC=1 bash scripts/r04/gpu_pmc.sh r05t11 > $out/pmc.log 2>&1; grep "fps_fat\|bq_grid_query" gpurun_out/pmc_r05t11/launches_of_one_pass.txt | cut -c1-110
