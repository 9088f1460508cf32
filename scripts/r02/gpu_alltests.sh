cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
( time timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1 | grep -v "^$"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
