cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["ms_per_step"], "cold", d["cold"]["scenes_per_s"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"], "blocked", d["host_blocked_frac"])'
B="--no-roofline --no-legs --cpu-scenes 0 --steps 64 --warmup 16"
python3 bench.py $B --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 2>&1 | python3 -c "$show" cfg5-65536
python3 bench.py $B --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 --scene beam 2>&1 | python3 -c "$show" cfg5-65536-beam
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
