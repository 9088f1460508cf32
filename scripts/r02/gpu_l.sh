cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_timed_path_gpu.py -m gpu -x -q 2>&1 | tail -4
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d.get("roofline",{}); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], "lat", d.get("latency",{}).get("ms_per_batch"), d["selfcheck"], "roof", r.get("frac"), r.get("launches_per_step"), r.get("kernel_ms_per_step"), [x for x in r.get("launches", []) if x[1]==1 and x[2] in (16640, 22784)])'
B="--no-legs --cpu-scenes 0 --worker"
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" wave
DET6D_NO_WAVE_CHAIN=1 python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" reg
python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" wave
DET6D_NO_WAVE_CHAIN=1 python3 bench.py --steps 192 --warmup 48 $B 2>/dev/null | python3 -c "$show" reg
python3 bench.py --steps 96 --warmup 16 $B --scene beam 2>/dev/null | python3 -c "$show" beam-wave
DET6D_NO_WAVE_CHAIN=1 python3 bench.py --steps 96 --warmup 16 $B --scene beam 2>/dev/null | python3 -c "$show" beam-reg
DET6D_DENSE_ROWS=1 python3 bench.py --steps 96 --warmup 16 $B --no-roofline 2>/dev/null | python3 -c "$show" dense-wave
DET6D_DENSE_ROWS=1 DET6D_NO_WAVE_CHAIN=1 python3 bench.py --steps 96 --warmup 16 $B --no-roofline 2>/dev/null | python3 -c "$show" dense-reg
