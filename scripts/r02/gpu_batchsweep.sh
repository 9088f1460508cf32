# how the rate moves with the scenes per pass (bench flags only; --batch != 8 is NOT BASELINE configs[1])
cd $GRAFT_REPO_ROOT
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 192 --warmup 48"
run() { python3 bench.py $B "$@" 2>/dev/null | python3 -c "$show" "$*"; }
run --streams 16 --prefetch 4 --group 4
run --batch 16 --streams 16 --prefetch 4 --group 4
run --batch 16 --streams 16 --prefetch 4 --group 2
run --batch 16 --streams 8 --prefetch 4 --group 2
run --batch 16 --streams 12 --prefetch 4 --group 2
run --batch 32 --streams 8 --prefetch 4 --group 1
run --batch 32 --streams 16 --prefetch 4 --group 1
run --batch 32 --streams 8 --prefetch 2 --group 2
run --batch 32 --streams 12 --prefetch 4 --group 1
run --batch 64 --streams 8 --prefetch 2 --group 1
