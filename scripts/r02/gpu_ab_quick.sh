# quick A/B of the default bench (no legs) on both scene generators, three runs each
cd $GRAFT_REPO_ROOT
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_mean"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2 3; do
python3 bench.py $B "$@" 2>/dev/null | python3 -c "$show" "uniform $*"
python3 bench.py $B --scene beam "$@" 2>/dev/null | python3 -c "$show" "beam $*"
done
