cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
timeout 2400 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 2400 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "regime or other_baseline or cooperative" 2>&1 | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], d['selfcheck'], d['latency_b1']['ms_per_frame'])"
