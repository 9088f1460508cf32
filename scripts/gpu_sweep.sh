cd $GRAFT_REPO_ROOT
python scripts/gpu_whatif.py base 22 2>&1 | tail -1
python scripts/gpu_whatif.py nofps 22 2>&1 | tail -1
python scripts/gpu_whatif.py nofps 6 2>&1 | tail -1
