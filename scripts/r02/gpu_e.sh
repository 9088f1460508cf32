cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_compact_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -8
