# full GPU suite on the current build, then A/B of the streaming form of the head's wide group (half-a-CU workgroups)
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
( time python3 -m pytest tests -x -q -m gpu ) 2>&1 | tail -6
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2; do
python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform"
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam"
DET6D_GROUP_STREAM=3 python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform stream3"
DET6D_GROUP_STREAM=3 python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam stream3"
done
