cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
timeout 900 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "full_size_vs" 2>&1 | grep -v "^    \|^$" | tail -40 | cut -c1-250
