cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --worker --no-roofline --steps 192 --warmup 48"
python3 bench.py $B 2>/dev/null | python3 -c "$show" base
DET6D_COMPACT_SPLIT=0 python3 bench.py $B 2>/dev/null | python3 -c "$show" split0-pow2-padding
DET6D_COMPACT_SPLIT=4 DET6D_COMPACT_SMIN=4 python3 bench.py $B 2>/dev/null | python3 -c "$show" granule4
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam-base
DET6D_COMPACT_SPLIT=0 python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam-split0
DET6D_COMPACT_SPLIT=4 DET6D_COMPACT_SMIN=4 python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam-granule4
