cd $GRAFT_REPO_ROOT
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --worker --no-roofline --steps 192 --warmup 48"
for q in 22 23 24 26 28; do GPU_MAX_HW_QUEUES=$q python3 bench.py $B 2>/dev/null | python3 -c "$show" queues$q; done
GPU_MAX_HW_QUEUES=24 python3 bench.py $B --streams 16 --sampler-streams 5 2>/dev/null | python3 -c "$show" q24-16+5
GPU_MAX_HW_QUEUES=24 python3 bench.py $B --streams 16 --sampler-streams 7 2>/dev/null | python3 -c "$show" q24-16+7
GPU_MAX_HW_QUEUES=24 python3 bench.py $B --streams 17 --sampler-streams 6 --group 4 2>/dev/null | python3 -c "$show" q24-17+6
