cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
NOPMC=1 bash scripts/r03/gpu_pmc.sh r03b_beam --scene beam
python3 -m pytest tests/test_timed_path_gpu.py -x -q -m gpu -k "two_rank" 2>&1 | tail -3
python3 -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "rows or tiny or fused" 2>&1 | tail -3
# pipeline shape sweep on beam
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"], d["latency_under_load"]["ms_p50_p99"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5 --scene beam"
python3 bench.py $B 2>/dev/null | python3 -c "$show" "beam default"
python3 bench.py $B --streams 12 2>/dev/null | python3 -c "$show" "beam streams12"
python3 bench.py $B --streams 8 2>/dev/null | python3 -c "$show" "beam streams8"
python3 bench.py $B --streams 12 --prefetch 2 --sampler-streams 3 2>/dev/null | python3 -c "$show" "beam streams12 pf2 ss3"
python3 bench.py $B --prefetch 2 2>/dev/null | python3 -c "$show" "beam pf2"
python3 bench.py $B --merge 2 2>/dev/null | python3 -c "$show" "beam merge2"
