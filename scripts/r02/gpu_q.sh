cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24 DET6D_EXPERIMENTS_LIB=1
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline",{}); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"], "roof", r.get("frac"), r.get("kernel_ms_per_step"))'
B="--no-legs --cpu-scenes 0 --worker --steps 192 --warmup 48"
for w in 0 8 4; do
DET6D_GROUP_WAVES=$w python3 bench.py $B 2>/dev/null | python3 -c "$show" waves$w
DET6D_GROUP_WAVES=$w python3 bench.py $B --scene beam --no-roofline 2>/dev/null | python3 -c "$show" beam-waves$w
done
