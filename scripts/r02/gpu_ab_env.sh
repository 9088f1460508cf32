# same-box A/B of one environment switch: $1 = variable, $2.. = values; three alternating rounds of the default bench (no legs)
cd $GRAFT_REPO_ROOT
var=$1; shift
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_mean"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2 3; do for v in "$@"; do
env $var=$v python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform $var=$v"
env $var=$v python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam $var=$v"
done; done
