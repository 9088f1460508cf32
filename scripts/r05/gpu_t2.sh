# round 5, second measurement: the whole GPU suite on the new query / NMS / S-FPS kernels, then same-box timing (uniform, ray-cast, 65536)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_t2; mkdir -p $out
( timeout 2400 python3 -m pytest tests -m gpu -q ) > $out/pytest_all.log 2>&1; tail -12 $out/pytest_all.log | cut -c1-200
export GPU_MAX_HW_QUEUES=24
for i in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/bench_$i.log 2> $out/bench_$i.err
grep '^{' $out/bench_$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('uniform', d['value'], d['selfcheck'], d['latency']['ms_per_batch'], d['latency_b1']['ms_per_frame'], d['latency_under_load']['ms_p50_p99'])"
done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --scene beam > $out/bench_beam.log 2> $out/bench_beam.err
grep '^{' $out/bench_beam.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam', d['value'], d['selfcheck'])"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 > $out/bench_65536.log 2> $out/bench_65536.err
grep '^{' $out/bench_65536.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('65536', d['value'], d['selfcheck'], d['latency_under_load']['ms_p50_p99'])"
STEPS=4 NOPMC=1 bash scripts/r04/gpu_pmc.sh r05t2 > $out/pmc.log 2>&1; grep "bq_grid\|compact\|fps_fat\|post_" gpurun_out/pmc_r05t2/launches_of_one_pass.txt
