cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "lat", d.get("latency",{}).get("ms_per_batch"))'
B="--no-legs --cpu-scenes 0 --worker --no-roofline --steps 192 --warmup 48"
for i in 1 2 3; do
DET6D_DENSE_ROWS=1 python3 bench.py $B 2>/dev/null | python3 -c "$show" dense-wave
DET6D_DENSE_ROWS=1 DET6D_NO_WAVE_CHAIN=1 python3 bench.py $B 2>/dev/null | python3 -c "$show" dense-reg
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam-wave
DET6D_NO_WAVE_CHAIN=1 python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam-reg
done
