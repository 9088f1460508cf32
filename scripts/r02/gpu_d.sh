cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
timeout 1200 python3 -m pytest tests/test_model_gpu.py tests/test_timed_path_gpu.py -m gpu -x -q 2>&1 | tail -5
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"], "blocked", d["host_blocked_frac"])'
B="--no-roofline --no-legs --cpu-scenes 0"
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" hoist
DET6D_NO_HOIST=1 python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" nohoist
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" hoist
DET6D_NO_HOIST=1 python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" nohoist
python3 bench.py --steps 192 --warmup 48 $B --streams 12 --prefetch 3 2>/dev/null | python3 -c "$show" hoist-12-3
python3 bench.py --steps 192 --warmup 48 $B --streams 12 --prefetch 4 2>/dev/null | python3 -c "$show" hoist-12-4
python3 bench.py --steps 192 --warmup 48 $B --streams 16 --prefetch 5 --sampler-streams 7 2>/dev/null | python3 -c "$show" hoist-16-5-7
python3 bench.py --steps 64 --warmup 16 $B --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 2>&1 | python3 -c "$show" cfg5-65536
python3 bench.py --steps 64 --warmup 16 $B --scene beam 2>&1 | python3 -c "$show" beam
