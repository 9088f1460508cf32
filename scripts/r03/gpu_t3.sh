# chain kernels (tags by shuffle, strided tiles): parity, per-launch table, A/B of pacing
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
python3 -m pytest tests/test_compact_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "chain or compact or mlp" 2>&1 | tail -4
python3 -m pytest tests/test_model_gpu.py tests/test_timed_path_gpu.py -x -q -m gpu -k "full_car or ray_cast or bench_group_full_size_vs_oracle or scene_pipeline" 2>&1 | tail -4
for sc in uniform beam; do
python3 bench.py --steps 96 --warmup 16 --worker --no-legs --cpu-scenes 0 --scene $sc 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$sc', d['value'], 'roof', r['frac'], r['kernel_ms_per_pass'], 'alg GF', r['algorithmic_gflop_per_pass'], 'sat', r['saturated'])
        for x in r['launches']: print(x)
"
done
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], d["config"]["window_ms_mean"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"], d["latency_under_load"]["ms_p50_p99"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2; do
for pace in 0 0.85 0.95; do
DET6D_PIPE_PACE=$pace python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform pace$pace"
DET6D_PIPE_PACE=$pace python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam pace$pace"
done
done
