cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_timed_path_gpu.py -m gpu -x -q 2>&1 | tail -3
bash scripts/r02/gpu_roof.sh
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --worker --no-roofline --steps 192 --warmup 48"
for i in 1 2; do
python3 bench.py $B 2>/dev/null | python3 -c "$show" uniform
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" beam
DET6D_DENSE_ROWS=1 python3 bench.py $B 2>/dev/null | python3 -c "$show" dense
done
