# round 5: tiles by ticket in the persistent group kernels — parity, then A/B against the static walk (experiments build), shipped numbers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t7; mkdir -p $out
( timeout 1500 python3 -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py tests/test_timed_path_gpu.py -m gpu -q -x ) > $out/pytest.log 2>&1; tail -5 $out/pytest.log | cut -c1-200
run() { tag=$1; shift
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], 'b8', d['latency']['ms_per_batch'], 'b1', d['latency_b1']['ms_per_frame'])" || tail -3 $out/b_$tag.err; }
export DET6D_EXPERIMENTS_LIB=1
DET6D_GROUP_STATIC=1 run exp_static
run exp_ticket
DET6D_GROUP_STATIC=1 run exp_static_beam --scene beam
run exp_ticket_beam --scene beam
unset DET6D_EXPERIMENTS_LIB
run ship
run ship2
run ship_beam --scene beam
run ship_65536 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8
STEPS=4 NOPMC=1 bash scripts/r04/gpu_pmc.sh r05t7 > $out/pmc.log 2>&1; grep "mlp_group" gpurun_out/pmc_r05t7/launches_of_one_pass.txt | cut -c1-110
