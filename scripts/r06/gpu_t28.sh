cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "weight" 2>&1 | tail -3
DET6D_EXPERIMENTS_LIB=1 DET6D_DBG_POISON_LDS=0x7F7F0000 timeout 1200 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "weighted_sampler" 2>&1 | tail -3
