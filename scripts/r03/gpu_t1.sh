# tests of the round's first batch of changes + quick A/B of the default bench on both scene generators
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
python3 -m pytest tests/test_ops_gpu.py tests/test_compact_gpu.py -x -q -m gpu -k "ball_query or compact" 2>&1 | tail -5
python3 -m pytest tests/test_golden_gpu.py tests/test_fp_gpu.py -x -q -m gpu 2>&1 | tail -5
python3 -m pytest tests/test_timed_path_gpu.py -x -q -m gpu -k "ray_cast or cooperative or sampler_failure or bench_entry" 2>&1 | tail -8
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], d["config"]["window_ms_mean"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"], d.get("latency_under_load"))'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2; do
python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform"
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam"
done
