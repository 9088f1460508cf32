cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --worker --no-roofline --steps 192 --warmup 48"
run() { python3 bench.py $B "$@" 2>/dev/null | python3 -c "$show" "$*"; }
run --streams 16 --sampler-streams 6 --group 4 --prefetch 4
run --streams 18 --sampler-streams 4 --group 3 --prefetch 4
run --streams 18 --sampler-streams 4 --group 6 --prefetch 2
run --streams 20 --sampler-streams 2 --group 4 --prefetch 4
run --streams 18 --sampler-streams 3 --group 3 --prefetch 5
run --streams 16 --sampler-streams 4 --group 4 --prefetch 4
run --streams 16 --sampler-streams 3 --group 4 --prefetch 4
run --streams 14 --sampler-streams 6 --group 2 --prefetch 6
run --streams 16 --sampler-streams 6 --group 4 --prefetch 4
