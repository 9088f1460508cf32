cd $GRAFT_REPO_ROOT
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["n_gpus"], d["value"], d["per_rank_scenes_per_s"], d["config"]["window_ms_min_median_max"], "stream_total", d["stream_total_s"], d["selfcheck"])'
export DET6D_BENCH_BACKEND=gloo
python3 bench.py --gpus 2 --steps 192 --warmup 48 2>/dev/null | python3 -c "$show" 2proc-long
python3 bench.py --gpus 2 --steps 960 --warmup 48 2>/dev/null | python3 -c "$show" 2proc-longer
python3 bench.py --gpus 1 --steps 960 --warmup 48 --no-legs --no-roofline --cpu-scenes 0 2>/dev/null | python3 -c "$show" 1proc-longer
GPU_MAX_HW_QUEUES=12 python3 bench.py --gpus 2 --steps 960 --warmup 48 --streams 8 --sampler-streams 3 2>/dev/null | python3 -c "$show" 2proc-8+3-longer
python3 bench.py --gpus 3 --steps 960 --warmup 48 2>/dev/null | python3 -c "$show" 3proc-longer
