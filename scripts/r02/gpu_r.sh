cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_timed_path_gpu.py -m gpu -x -q 2>&1 | tail -3
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline",{}); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"], "roof", r.get("frac"), r.get("launches_per_step"), r.get("kernel_ms_per_step"))'
B="--no-legs --cpu-scenes 0 --worker --steps 192 --warmup 48"
for i in 1 2 3; do python3 bench.py $B 2>/dev/null | python3 -c "$show" sums-in-rows; done
