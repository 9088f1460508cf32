# round 6: sequencers with deferred publish, final form: sampler suites, the parity suites that run the samplers in the model
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
python3 tests/gpu_scripts/fps_coop.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 2400 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 2400 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "regime or other_baseline or cooperative or full_size_vs" 2>&1 | tail -3
