cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_dbg1; mkdir -p $out
timeout 600 python3 scripts/r05/dbg_bq.py 2>&1 | grep -v amdgpu.ids | head -4
( timeout 2400 python3 -m pytest tests -m gpu -q -x ) > $out/pytest_all.log 2>&1; tail -12 $out/pytest_all.log | cut -c1-200
