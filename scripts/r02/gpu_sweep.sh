cd $GRAFT_REPO_ROOT
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 192 --warmup 48"
run() { q=$1; shift; GPU_MAX_HW_QUEUES=$q python3 bench.py $B "$@" 2>/dev/null | python3 -c "$show" "q$q $*"; }
run 24 --streams 16 --prefetch 4 --sampler-streams 6 --group 4
run 24 --streams 16 --prefetch 3 --sampler-streams 6 --group 4
run 24 --streams 16 --prefetch 5 --sampler-streams 6 --group 4
run 24 --streams 18 --prefetch 4 --sampler-streams 5 --group 3
run 24 --streams 16 --prefetch 2 --sampler-streams 6 --group 8
run 24 --streams 16 --prefetch 8 --sampler-streams 6 --group 2
run 24 --streams 20 --prefetch 4 --sampler-streams 4 --group 4
run 32 --streams 20 --prefetch 4 --sampler-streams 6 --group 4
run 32 --streams 24 --prefetch 4 --sampler-streams 6 --group 4
run 16 --streams 12 --prefetch 4 --sampler-streams 4 --group 4
run 24 --streams 16 --prefetch 4 --sampler-streams 4 --group 4
run 24 --streams 16 --prefetch 4 --sampler-streams 8 --group 4
run 24 --streams 16 --prefetch 4 --sampler-streams 6 --group 4 --batch 16
