cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_t13; mkdir -p $out
( timeout 1500 python3 -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "cooperative or weighted or fps" ) > $out/pytest.log 2>&1; tail -6 $out/pytest.log | cut -c1-220
