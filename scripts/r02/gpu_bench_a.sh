# first GPU check of round 2: the driver's bench command vs the long run, the 2-rank entry point (dry run on one GPU), GPU tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02a
export GPU_MAX_HW_QUEUES=24
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r02a/bench_20.log 2>&1
tail -4 gpurun_out/r02a/bench_20.log | cut -c1-1500
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-roofline --no-legs --cpu-scenes 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('short', d['value'], d['cold']['scenes_per_s'], d['latency'], d['selfcheck'])"; done
python3 bench.py --gpus 1 --steps 192 --warmup 48 --no-roofline --no-legs --cpu-scenes 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('long', d['value'], d['cold']['scenes_per_s'], d['selfcheck'])"
python3 bench.py --gpus 1 --steps 192 --warmup 48 --no-roofline --no-legs --cpu-scenes 0 --distinct-batches 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('long-1batch', d['value'], d['cold']['scenes_per_s'], d['selfcheck'])"
python3 bench.py --gpus 1 --steps 192 --warmup 48 --no-roofline --no-legs --cpu-scenes 0 --scene beam 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('long-beam', d['value'], d['cold']['scenes_per_s'], d['selfcheck'], d['compact_fill'])"
DET6D_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r02a/bench_2ranks.log 2>&1; echo rc=$?
tail -2 gpurun_out/r02a/bench_2ranks.log | cut -c1-1200
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
