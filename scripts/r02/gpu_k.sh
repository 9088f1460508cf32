cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline",{}); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --worker --no-roofline"
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" base
DET6D_FORK_GROUPS=1 python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" fork
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" base
DET6D_FORK_GROUPS=1 python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" fork
DET6D_FORK_GROUPS=1 python3 bench.py --steps 192 --warmup 48 $B --streams 12 --sampler-streams 5 2>/dev/null | python3 -c "$show" fork-12
DET6D_FORK_GROUPS=1 python3 bench.py --steps 96 --warmup 16 $B --scene beam 2>/dev/null | python3 -c "$show" beam-fork
python3 bench.py --steps 96 --warmup 16 $B --scene beam 2>/dev/null | python3 -c "$show" beam-base
