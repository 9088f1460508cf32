cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02b
export GPU_MAX_HW_QUEUES=24
timeout 1500 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q 2>&1 | tail -15
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["n_gpus"], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], "blocked", d["host_blocked_frac"], d["selfcheck"])'
B="--no-roofline --no-legs --cpu-scenes 0"
for i in 1 2 3 4 5; do python3 bench.py --steps 20 --warmup 5 $B 2>/dev/null | python3 -c "$show" short; done
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" long
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" long
