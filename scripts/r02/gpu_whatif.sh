cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], d["host_blocked_frac"])'
B="--no-legs --cpu-scenes 0 --worker --no-roofline --steps 192 --warmup 48 --no-selfcheck"
export DET6D_EXPERIMENTS_LIB=1
for i in 1 2; do
python3 bench.py $B 2>/dev/null | python3 -c "$show" base
DET6D_FPS_STANDIN=3 python3 bench.py $B 2>&1 | python3 -c "$show" free-SA1-dfps
DET6D_FPS_STANDIN=4 python3 bench.py $B 2>&1 | python3 -c "$show" free-all-samplers
done
DET6D_FPS_STANDIN=4 python3 bench.py $B --scene beam 2>&1 | python3 -c "$show" beam-free-all-samplers
python3 bench.py $B --scene beam 2>&1 | python3 -c "$show" beam-base
